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


# ------------------------------------------------------------------------------------------------------------
# the attacks batched over images.  Every image of a batch must come out as from its own one-image run
# (which the goldens above pin to the reference), whatever the other images of the batch do: per-sample norms, masks, early
# exits, step sizes, bests.
def _content_noise(image):
    """N(0,1) noise as a function of the image itself, so that batched and one-image runs draw the same noise per image"""
    out = []
    for b in range(image.shape[0]):
        g = torch.Generator().manual_seed(int(image[b].double().sum().item() * 1e6) % (2 ** 31))
        out.append(torch.randn(image[b:b + 1].shape, generator=g))
    return torch.cat(out, dim=0)


@pytest.mark.parametrize('name', ['fgsm', 'deepfool', 'apgd_ce', 'apgd_dlr', 'fab', 'cw'])
def test_batched_attack_equals_the_one_image_runs(name, gold, monkeypatch):
    from gen_adversarial_amd.attacks import l2_attacks
    monkeypatch.setattr(l2_attacks, '_per_image_randn', _content_noise)
    net = toy_net()
    images, labels = torch.from_numpy(gold['images']), torch.from_numpy(gold['labels'])
    # a fourth image that is misclassified from the start (returns success with distortion 0 and must not disturb the others)
    images = torch.cat([images, images[:1].flip(-1)], dim=0)
    with torch.no_grad():
        wrong = (net(images[3:]).argmax(dim=1) + 1) % 6
    labels = torch.cat([labels, wrong], dim=0)
    singles = [ATTACKS[name]()(images[i:i + 1].clone(), labels[i:i + 1].clone(), net) for i in range(4)]
    s, b, a = ATTACKS[name]()(images.clone(), labels.clone(), net)
    assert s.shape == (4,) and b.shape == (4,) and a.shape == images.shape
    for i, (s1, b1, a1) in enumerate(singles):
        assert bool(s[i]) == bool(s1), (name, i)
        if bool(s1):
            assert float(b[i]) == pytest.approx(float(b1), rel=1e-5, abs=1e-6), (name, i)
        np.testing.assert_allclose(a[i:i + 1].detach().numpy(), a1.detach().numpy(), atol=1e-5, err_msg=f'{name} {i}')
    assert bool(s[3]) and float(b[3]) == 0.0 or name.startswith('apgd') or name == 'cw'    # APGD / C&W have no "already misclassified" exit


def test_batched_autoattack_equals_the_one_image_runs(gold, monkeypatch):
    from gen_adversarial_amd.attacks import l2_attacks
    monkeypatch.setattr(l2_attacks, '_per_image_randn', _content_noise)
    net = toy_net()
    images, labels = torch.from_numpy(gold['images']), torch.from_numpy(gold['labels'])

    def make():
        aa = AutoAttack()
        for atk in aa.ce + aa.dlr:
            atk.__init__(n_iter=10, rho=0.75, max_bound=atk.max_bound, ce_loss=not (atk.criterion == atk.dlr_loss))
        aa.fab.n_iter = 6
        return aa
    singles = [make()(images[i:i + 1].clone(), labels[i:i + 1].clone(), net) for i in range(3)]
    s, b, a = make()(images.clone(), labels.clone(), net)
    for i, (s1, b1, a1) in enumerate(singles):
        assert bool(s[i]) == bool(s1)
        assert float(b[i]) == pytest.approx(float(b1), rel=1e-5, abs=1e-6)
        np.testing.assert_allclose(a[i:i + 1].numpy(), a1.numpy(), atol=1e-5)


def test_class_gradients_one_pass_per_class_for_the_whole_batch():
    """the multi-class VJP of DeepFool / FAB: per image the gradients of ITS OWN class list, from one forward"""
    from gen_adversarial_amd.attacks.l2_attacks import class_gradients
    net = toy_net()
    x = torch.rand(3, 3, 16, 16, generator=torch.Generator().manual_seed(2))
    classes = torch.tensor([[0, 3], [5, 1], [2, 2]])
    y, g = class_gradients(net, x, classes)
    assert g.shape == (3, 2, 3, 16, 16)
    for b in range(3):
        for j in range(2):
            xr = x[b:b + 1].clone().requires_grad_(True)
            (ref,) = torch.autograd.grad(net(xr)[0, classes[b, j]], [xr])
            np.testing.assert_allclose(g[b, j].numpy(), ref[0].numpy(), atol=1e-6)
    y2, g2 = class_gradients(net, x)
    assert g2.shape == (3, 6, 3, 16, 16) and torch.allclose(g2[1, 5], g[1, 0], atol=1e-6)


def test_class_jacobian_asks_the_defender_first_and_deepfool_pays_for_gradients_only_while_active(gold):
    """ClassJacobian protocol: a net that offers `class_jacobian(x, classes)` (the HIP defender's K-cotangent
    plan) is asked instead of one autograd backward per class; DeepFool reads the logits first and requests the K gradients only
    while some image is still active (the terminating iteration is a forward alone)."""
    from gen_adversarial_amd.attacks.l2_attacks import ClassJacobian
    net = toy_net()
    stats = {'forwards': 0, 'grads': 0}

    class Offer:
        def __init__(self, x, classes):
            stats['forwards'] += 1
            self.inner = None
            self.x, self.classes = x, classes
            with torch.no_grad():
                self.logits = net(x)

        def grads(self):
            stats['grads'] += 1
            net.class_jacobian = None          # the inner ClassJacobian must take the autograd path
            try:
                return ClassJacobian(net, self.x, self.classes).grads()
            finally:
                net.class_jacobian = offer

    def offer(x, classes=None):
        return Offer(x, classes)
    net.class_jacobian = offer
    try:
        images, labels = torch.from_numpy(gold['images']), torch.from_numpy(gold['labels'])
        s, b, a = DeepFool(num_classes=4, overshoot=0.02, max_iter=10)(images.clone(), labels.clone(), net)
        # the results are those of the plain path (the goldens of the reference's own DeepFool cover that one)
        del net.class_jacobian
        s0, b0, a0 = DeepFool(num_classes=4, overshoot=0.02, max_iter=10)(images.clone(), labels.clone(), net)
        assert torch.equal(torch.as_tensor(s), torch.as_tensor(s0)) and torch.allclose(torch.as_tensor(b), torch.as_tensor(b0))
        np.testing.assert_allclose(a.numpy(), a0.numpy(), atol=1e-6)
        assert stats['forwards'] >= 2 and stats['grads'] == stats['forwards'] - 1, stats     # the last iteration asked for no gradient
    finally:
        if hasattr(net, 'class_jacobian'):
            del net.class_jacobian
