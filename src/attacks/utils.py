from gen_adversarial_amd.attacks.utils import l2_norm, normalize, projection_l2  # noqa: F401
