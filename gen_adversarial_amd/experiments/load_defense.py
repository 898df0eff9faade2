"""
Config / factory layer with the reference's surface (src/experiments/load_defense.py:17-146):
`load(args) -> (args, defense_model)`; reads the same flat yaml keys (`classifier_path`, `autoencoder_path`,
`interpolation_alphas`, `alpha_attenuation`, `initial_noise_eps`, `gaussian_blur_input`), sets `args.image_size`,
`args.attacks` and attaches `defense_model.get_purified`.

Built: every defense_type of the reference for every experiment — 'base' | 'trades' (classifier only), 'ablation' (noise /
blur), 'ours' (ids: NVAE; gender: e4e + StyleGAN2; cars: Style-Transformer + StyleGAN2) and the competitors 'A-VAE' / 'ND-VAE'
(load_defense.py:95-124) — plus the reference's attack sets (DeepFool, C&W, AutoAttack) and `args.pgd` (PGD-Linf).
An unknown experiment or defense type raises NotImplementedError, like in the reference (:75,:144).
"""
from argparse import Namespace

import yaml

from ..attacks.l2_attacks import AutoAttack, CW, DeepFool
from ..attacks.pgd import PGDLinf
from ..defenses.ablations.models import GaussianBlurDefenseModel, GaussianNoiseDefenseModel
from ..defenses.competitors.a_vae import AVaeDefenseModel, load_AVAE
from ..defenses.competitors.nd_vae import NDVaeDefenseModel, load_NDVAE
from ..defenses.ours.models import (CarsTypeClassifier, CelebaGenderClassifier, CelebaIdentityClassifier,
                                   E4EStyleGanDefenseModel, NVAEDefenseModel, TransStyleGanDefenseModel)
from ..defenses.wrappers import EoTWrapper


def load(args: Namespace):
    with open(args.config, 'r', encoding='utf-8') as stream:
        d_params = Namespace(**yaml.safe_load(stream))

    if args.experiment == 'ids':
        args.image_size = 64
        # the reference's evaluation attacks with its hyper-parameters for this experiment (load_defense.py:48-52)
        args.attacks = {
            'deepfool': DeepFool(num_classes=8, overshoot=0.02, max_iter=128),
            'c&w': CW(c=16., kappa=0.05, steps=1024, lr=5e-3, n_restarts=8),
            'autoattack': AutoAttack()
        }
        # PGD-Linf eps=8/255 is the attack BASELINE.json names (not in the reference tree); same call protocol,
        # selected with `--attack pgd`
        args.pgd = PGDLinf(eps=8.0 / 255.0, step_size=2.0 / 255.0, steps=40)
        args.pgd_bpda = PGDLinf(eps=8.0 / 255.0, step_size=2.0 / 255.0, steps=40, bpda=True)     # BASELINE.json configs[3]
        base_classifier = CelebaIdentityClassifier(d_params.classifier_path, args.device)
        hl_instance = NVAEDefenseModel
    elif args.experiment == 'gender':
        # ResNet-50 classifier + e4e / StyleGAN2 purifier (load_defense.py:27-41)
        args.image_size = 256
        args.attacks = {
            'deepfool': DeepFool(num_classes=2, overshoot=0.01, max_iter=1024),
            'c&w': CW(c=64., kappa=0.01, steps=1024, lr=1e-3, n_restarts=8, early_stopping_steps=32),
            'autoattack': AutoAttack()
        }
        args.pgd = PGDLinf(eps=8.0 / 255.0, step_size=2.0 / 255.0, steps=40)
        base_classifier = CelebaGenderClassifier(d_params.classifier_path, args.device)
        hl_instance = E4EStyleGanDefenseModel
    elif args.experiment == 'cars':
        # ResNeXt-50 classifier + Style-Transformer / StyleGAN2 purifier (load_defense.py:59-73)
        args.image_size = 128
        args.attacks = {
            'deepfool': DeepFool(num_classes=4, overshoot=0.02, max_iter=256),
            'c&w': CW(c=24., kappa=0.02, steps=1024, lr=2e-3, n_restarts=8),
            'autoattack': AutoAttack()
        }
        args.pgd = PGDLinf(eps=8.0 / 255.0, step_size=2.0 / 255.0, steps=40)
        base_classifier = CarsTypeClassifier(d_params.classifier_path, args.device)
        hl_instance = TransStyleGanDefenseModel
    else:
        raise NotImplementedError

    if args.defense_type in ('base', 'trades'):
        defense_model = base_classifier
        defense_model.get_purified = lambda x: x
    elif args.defense_type == 'ablation':
        if d_params.type == 'noise':
            defense_model = GaussianNoiseDefenseModel(base_classifier, 2. if args.experiment == 'ids' else 4.)
        else:
            defense_model = GaussianBlurDefenseModel(base_classifier)
        defense_model = EoTWrapper(defense_model, args.eot_steps)
        defense_model.get_purified = lambda x: defense_model.model.purify(x)
    elif args.defense_type == 'A-VAE':
        # load_defense.py:95-106: StyledGenerator(args.image_size), AVaeDefenseModel(base_classifier, a_vae, kernel_size)
        a_vae = load_AVAE(d_params.autoencoder_path, args.image_size)
        defense_model = AVaeDefenseModel(base_classifier, a_vae, d_params.kernel_size)
        defense_model = EoTWrapper(defense_model, args.eot_steps)
        defense_model.get_purified = lambda x: defense_model.model.purify(x)
    elif args.defense_type == 'ND-VAE':
        # load_defense.py:108-124: Defence_NVAE(x_channels, encoding_channels, pre_proc_groups, scales, groups, cells, image_size)
        nd_vae = load_NDVAE(d_params.autoencoder_path, d_params.x_channels, d_params.encoding_channels, d_params.pre_proc_groups,
                            d_params.scales, d_params.groups, d_params.cells, args.image_size)
        defense_model = NDVaeDefenseModel(base_classifier, nd_vae, d_params.noise_std)
        defense_model = EoTWrapper(defense_model, args.eot_steps)
        defense_model.get_purified = lambda x: defense_model.model.purify(x)
    elif args.defense_type == 'ours':
        defense_model = hl_instance(base_classifier, d_params.autoencoder_path, d_params.interpolation_alphas,
                                    d_params.alpha_attenuation, d_params.initial_noise_eps,
                                    d_params.gaussian_blur_input, args.device)
        defense_model = EoTWrapper(defense_model, args.eot_steps)
        defense_model.get_purified = lambda x: defense_model.model(x, preds_only=False)[-1]
    else:
        raise NotImplementedError
    return args, defense_model
