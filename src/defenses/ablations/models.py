from gen_adversarial_amd.defenses.ablations.models import GaussianBlurDefenseModel, GaussianNoiseDefenseModel  # noqa: F401
