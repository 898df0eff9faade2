from gen_adversarial_amd.defenses.competitors.a_vae import AVaeDefenseModel, load_AVAE  # noqa: F401
