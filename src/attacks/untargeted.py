from gen_adversarial_amd.attacks.l2_attacks import (APGDAttack, AutoAttack, CW, DeepFool, FABAttack, FGSM,  # noqa: F401
                                                    UntargetedL2Attack)
from gen_adversarial_amd.attacks.pgd import PGDLinf  # noqa: F401
