from gen_adversarial_amd.defenses.competitors.nd_vae import NDVaeDefenseModel, load_NDVAE  # noqa: F401
