"""
Golden vectors for the L2 attacks: runs the REFERENCE attack classes (src/attacks/untargeted.py, imported read-only from
/root/reference — pure torch/numpy, no shims needed) on a small seeded conv net on the CPU and records
(success, L2, adversarial image) under fixed torch seeds.  Container-only; the .npz travels, the reference does not.

    python tests/golden/make_attack_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path = [p_ for p_ in sys.path if os.path.abspath(p_ or '.') != os.path.dirname(os.path.dirname(HERE))]
sys.path.insert(0, '/root/reference')        # the reference's `src` package, not this repo's import shim

from src.attacks.untargeted import APGDAttack, AutoAttack, CW, DeepFool, FABAttack, FGSM   # noqa: E402
from src.attacks.utils import projection_l2   # noqa: E402


def toy_net(seed=0, n_classes=6):
    torch.manual_seed(seed)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.SiLU(), torch.nn.AvgPool2d(2),
                              torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.SiLU(), torch.nn.Flatten(),
                              torch.nn.Linear(8 * 8 * 8, n_classes))
    return net.eval()


def main():
    net = toy_net()
    g = torch.Generator().manual_seed(7)
    out = {}
    images = torch.rand(3, 3, 16, 16, generator=g)
    out['images'] = images.numpy()
    with torch.no_grad():
        labels = net(images).argmax(dim=1)
    out['labels'] = labels.numpy()
    attacks = {
        'fgsm': lambda: FGSM(l2_bound=0.5),
        'deepfool': lambda: DeepFool(num_classes=5, overshoot=0.02, max_iter=50),
        'cw': lambda: CW(c=64., kappa=0.05, steps=120, lr=1e-2, n_restarts=3, early_stopping_steps=8),
        'apgd_ce': lambda: APGDAttack(n_iter=20, rho=0.75, max_bound=0.5, ce_loss=True),
        'apgd_dlr': lambda: APGDAttack(n_iter=20, rho=0.75, max_bound=1.0, ce_loss=False),
        'fab': lambda: FABAttack(n_iter=12, alpha_max=0.1, eta=1.05, beta=0.9),
    }
    for name, mk in attacks.items():
        for i in range(images.shape[0]):
            torch.manual_seed(100 + i)
            s, b, a = mk()(images[i:i + 1].clone(), labels[i:i + 1].clone(), net)
            out[f'{name}_{i}_success'] = np.asarray(bool(s))
            out[f'{name}_{i}_bound'] = np.asarray(float(b))
            out[f'{name}_{i}_adv'] = a.detach().numpy()
            print(name, i, bool(s), float(b))
    # the composite (shortened budgets to keep the fixture generation fast: same classes, fewer iterations)
    aa = AutoAttack()
    for atk in (aa.apgd_ce1, aa.apgd_ce2, aa.apgd_ce3, aa.apgd_dlr1, aa.apgd_dlr2, aa.apgd_dlr3):
        atk.__init__(n_iter=10, rho=0.75, max_bound=atk.max_bound, ce_loss=not (atk.criterion == atk.dlr_loss))
    aa.fab.n_iter = 6
    torch.manual_seed(321)
    s, b, a = aa(images[:1].clone(), labels[:1].clone(), net)
    out['aa_success'], out['aa_bound'], out['aa_adv'] = np.asarray(bool(s)), np.asarray(float(b)), a.detach().numpy()
    print('autoattack', bool(s), float(b))
    # projection_l2 on random rows
    p = torch.rand(5, 40, generator=g)
    w = torch.randn(5, 40, generator=g)
    w[0, :5] = 0.0
    b_ = torch.randn(5, 1, generator=g)
    out['proj_p'], out['proj_w'], out['proj_b'] = p.numpy(), w.numpy(), b_.numpy()
    out['proj_d'] = projection_l2(p, w, b_).numpy()
    np.savez_compressed(os.path.join(HERE, 'attacks_toy.npz'), **out)


if __name__ == '__main__':
    main()
