from gen_adversarial_amd.defenses.ours.abstract_models import BaseClassificationModel, MLVGMDefenseModel  # noqa: F401
