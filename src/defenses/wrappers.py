from gen_adversarial_amd.defenses.wrappers import EoTWrapper  # noqa: F401
