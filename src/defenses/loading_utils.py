from gen_adversarial_amd.defenses.loading_utils import *  # noqa: F401,F403
from gen_adversarial_amd.defenses.loading_utils import load_Vgg11, load_NVAE, load_ResNet50, load_ResNext50, load_E4EStyleGan, load_TranStyleGan  # noqa: F401
