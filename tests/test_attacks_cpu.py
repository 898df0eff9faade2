"""
CPU: our restatement of the reference's L2 attacks against golden vectors produced by the reference's own attack
classes on a toy net under fixed seeds (tests/golden/make_attack_golden.py).  Same seeds, same draws => same results.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from gen_adversarial_amd.attacks.l2_attacks import APGDAttack, AutoAttack, CW, DeepFool, FABAttack, FGSM
from gen_adversarial_amd.attacks.utils import l2_norm, normalize, projection_l2


def toy_net(seed=0, n_classes=6):
    torch.manual_seed(seed)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.SiLU(), torch.nn.AvgPool2d(2),
                              torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.SiLU(), torch.nn.Flatten(),
                              torch.nn.Linear(8 * 8 * 8, n_classes))
    return net.eval()


ATTACKS = {
    'fgsm': lambda: FGSM(l2_bound=0.5),
    'deepfool': lambda: DeepFool(num_classes=5, overshoot=0.02, max_iter=50),
    'cw': lambda: CW(c=64., kappa=0.05, steps=120, lr=1e-2, n_restarts=3, early_stopping_steps=8),
    'apgd_ce': lambda: APGDAttack(n_iter=20, rho=0.75, max_bound=0.5, ce_loss=True),
    'apgd_dlr': lambda: APGDAttack(n_iter=20, rho=0.75, max_bound=1.0, ce_loss=False),
    'fab': lambda: FABAttack(n_iter=12, alpha_max=0.1, eta=1.05, beta=0.9),
}


@pytest.fixture(scope='module')
def gold():
    return load_golden('attacks_toy.npz')


@pytest.mark.parametrize('name', list(ATTACKS))
def test_attack_matches_reference(name, gold):
    net = toy_net()
    images, labels = torch.from_numpy(gold['images']), torch.from_numpy(gold['labels'])
    for i in range(images.shape[0]):
        torch.manual_seed(100 + i)
        s, b, a = ATTACKS[name]()(images[i:i + 1].clone(), labels[i:i + 1].clone(), net)
        assert bool(s) == bool(gold[f'{name}_{i}_success']), (name, i)
        assert float(b) == pytest.approx(float(gold[f'{name}_{i}_bound']), rel=1e-4, abs=1e-6), (name, i)
        np.testing.assert_allclose(a.detach().numpy(), gold[f'{name}_{i}_adv'], atol=2e-5, err_msg=f'{name} {i}')


def test_autoattack_matches_reference(gold):
    net = toy_net()
    images, labels = torch.from_numpy(gold['images']), torch.from_numpy(gold['labels'])
    aa = AutoAttack()
    for atk in aa.ce + aa.dlr:
        atk.__init__(n_iter=10, rho=0.75, max_bound=atk.max_bound, ce_loss=not (atk.criterion == atk.dlr_loss))
    aa.fab.n_iter = 6
    torch.manual_seed(321)
    s, b, a = aa(images[:1].clone(), labels[:1].clone(), net)
    assert bool(s) == bool(gold['aa_success'])
    assert float(b) == pytest.approx(float(gold['aa_bound']), rel=1e-4)
    np.testing.assert_allclose(a.numpy(), gold['aa_adv'], atol=2e-5)


def test_projection_l2_matches_reference_and_is_feasible(gold):
    p, w, b = (torch.from_numpy(gold[k]) for k in ('proj_p', 'proj_w', 'proj_b'))
    d = projection_l2(p, w, b)
    np.testing.assert_allclose(d.numpy(), gold['proj_d'], atol=1e-6)
    q = p + d
    assert float(q.min()) >= -1e-5 and float(q.max()) <= 1 + 1e-5           # stays in the box


def test_norm_helpers_are_per_sample():
    x = torch.randn(3, 2, 4, 4)
    n = l2_norm(x)
    assert n.shape == (3,) and torch.allclose(n[1], x[1].flatten().norm())
    assert torch.allclose(l2_norm(normalize(x)), torch.ones(3))
    assert l2_norm(x, keepdim=True).shape == (3, 1, 1, 1)


def test_dlr_needs_four_classes():
    atk = APGDAttack(n_iter=2, rho=0.75, max_bound=0.5, ce_loss=False)
    with pytest.raises(AttributeError):
        atk.dlr_loss(torch.randn(1, 3), torch.tensor([0]))
