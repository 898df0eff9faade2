from gen_adversarial_amd.experiments.load_defense import load  # noqa: F401
