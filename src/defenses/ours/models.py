from gen_adversarial_amd.defenses.ours.models import (CelebaGenderClassifier, CelebaIdentityClassifier, CarsTypeClassifier,  # noqa: F401
                                                      E4EStyleGanDefenseModel, NVAEDefenseModel, TransStyleGanDefenseModel)
