from gen_adversarial_amd.attacks.pgd import PGDLinf  # noqa: F401
