"""
The reference's untargeted L2 evaluation attacks, restated (src/attacks/untargeted.py): same class names, constructor
arguments and call protocol `attack(image (1,3,H,W), gt_label (1,), net) -> (success: bool, l2: float, adv)`, same update
rules and constants, so that `results.json` keeps its meaning (distortion per image, failure encoded by the driver).

    FGSM        untargeted.py:708-750      one L2-normalised sign step (initialiser of C&W)
    DeepFool    untargeted.py:470-568      closest linearised decision boundary among the top-k classes
    CW          untargeted.py:325-467      Carlini-Wagner L2 in tanh space, Adam, restarts with adaptive c
    APGDAttack  untargeted.py:37-243       Auto-PGD (CE or DLR loss) with momentum and step-size halving
    FABAttack   untargeted.py:571-705      Fast Adaptive Boundary attack
    AutoAttack  untargeted.py:246-322      APGD-CE x3 bounds -> APGD-DLR x3 bounds -> FAB, keep the smallest success

What differs from the reference, deliberately:
  * everything stays on the device of `image` (no numpy round trips: DeepFool's float64/float32 mix is reproduced with
    torch dtypes), one `.item()`-style synchronisation per decision instead of per tensor op;
  * gradients are taken with `torch.autograd.grad(..., [x])`, i.e. dX only — the reference's `.backward()` call sites also
    accumulate weight gradients that nobody reads;
  * DeepFool / FAB need one input-gradient per class on the SAME forward: they go through `class_gradients`, which issues
    the vector-Jacobian products back to back on one retained graph (the HIP defender replays its backward plan per call).
  * DeepFool, FAB, APGD, AutoAttack, C&W and FGSM are BATCHED over images (`batched = True`): `image` may be
    (B,3,H,W) with `gt_label` (B,), every image attacked independently — per-sample norms, masks, early exits, step sizes and
    bests — while every defender call carries B x EoT rows and every per-class backward pass serves all B images at once
    (one vector-Jacobian pass per class RANK for the whole batch: untargeted.py:526-560, :605-635 issue them per image).  For
    B = 1 the arithmetic is the reference's, operation for operation (tests/golden/attacks_toy.npz); for B > 1 every image's
    result equals its own B = 1 run (tests/test_attacks_cpu.py).  C&W is batched the same way (per-image Adam state is
    elementwise; the gradient clip, the early stop and the adaptive c are per image).
Random draws use `torch.randn_like` per image in the reference's order, so a CPU run under `torch.manual_seed` reproduces
the reference bit for bit (tests/golden/attacks_*.npz).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import List, Tuple

import numpy as np
import torch
from torch import nn

from .utils import l2_norm, normalize, projection_l2


class UntargetedL2Attack(ABC):
    """call protocol of untargeted.py:13-34"""

    @abstractmethod
    def __call__(self, image: torch.Tensor, gt_label: torch.Tensor, net: nn.Module) -> Tuple[bool, float, torch.Tensor]:
        ...


class ClassJacobian:
    """ONE forward of the whole batch now (`logits`, (B, n)), the per-class input gradients d logits[b, k_b] / d x_b LATER and only
    on demand (`grads()`, (B, K, *x.shape[1:])): an attack decides from the logits whether any image is still active before it
    pays for the K backward passes (DeepFool's terminating iteration needs none).
    classes: None (all n classes, the same for every image) or a (B, K) index tensor (each image its own class list).  One
    backward pass per class COLUMN serves all B images (independent rows of the defender); a defender that offers
    `class_jacobian` (the HIP engine's K-cotangent backward plan: several columns per replay) is asked first.  With a stochastic
    defender every gradient belongs to the same draw."""

    def __init__(self, net: nn.Module, x: torch.Tensor, classes=None):
        self.classes = classes
        fast = getattr(net, 'class_jacobian', None)
        self._fast = fast(x.detach(), classes) if fast is not None else None       # object with .logits / .grads(), or None
        if self._fast is not None:
            self.logits = self._fast.logits
            return
        self.x = x.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            self.y = net(self.x)
        self.logits = self.y.detach()

    def grads(self) -> torch.Tensor:
        if self._fast is not None:
            return self._fast.grads()
        y, classes = self.y, self.classes
        with torch.enable_grad():
            if classes is None:
                cols = [y[:, k].sum() for k in range(y.shape[1])]
            else:
                cols = [y.gather(1, classes[:, j:j + 1]).sum() for j in range(classes.shape[1])]
            grads = [torch.autograd.grad(c, [self.x], retain_graph=True)[0] for c in cols]
        return torch.stack(grads, dim=1)


def class_gradients(net: nn.Module, x: torch.Tensor, classes=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """logits (B, n) and d logits[b, k_b] / d x_b (B, K, *x.shape[1:]) of one forward (see ClassJacobian)"""
    j = ClassJacobian(net, x, classes)
    return j.logits, j.grads()


def _per_image_randn(image: torch.Tensor) -> torch.Tensor:
    """N(0,1) noise drawn image by image (the order B one-image attacks would draw it in)"""
    return torch.cat([torch.randn_like(image[b:b + 1]) for b in range(image.shape[0])], dim=0)


def _ret(success: torch.Tensor, bound: torch.Tensor, adv: torch.Tensor):
    """the protocol's (bool, float, adv) for one image; tensors (B,), (B,), (B,...) for a batch"""
    if adv.shape[0] == 1:
        return bool(success.item()), float(bound.item()), adv
    return success, bound, adv


def _bc(v: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    return v.view(-1, *([1] * (like.dim() - 1)))


# ---------------------------------------------------------------------------------------------------------------------
class FGSM(UntargetedL2Attack):
    batched = True

    def __init__(self, l2_bound: float):
        self.l2_bound = l2_bound

    def __call__(self, image, gt_label, net):
        x = image.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            logits = net(x)
            already = torch.argmax(logits, dim=-1) != gt_label.view(-1)
            if bool(already.all()):
                return _ret(already, torch.zeros(x.shape[0], device=x.device), image)
            cost = -nn.functional.cross_entropy(logits, gt_label.view(-1), reduction='sum')     # per-image gradients: rows are independent
            (g,) = torch.autograd.grad(cost, [x])
        step = g.sign()
        step = step / torch.norm(step.view(step.size(0), -1), p=2, dim=1, keepdim=True).view(-1, 1, 1, 1)
        x_adv = torch.clamp(x.detach() - step * self.l2_bound, 0., 1.)
        with torch.no_grad():
            fooled = torch.argmax(net(x_adv), dim=-1) != gt_label.view(-1)
        x_adv = torch.where(_bc(already, x_adv), image.detach(), x_adv)
        bound = torch.where(already, torch.zeros_like(fooled, dtype=torch.float32), torch.full_like(fooled, self.l2_bound, dtype=torch.float32))
        return _ret(fooled | already, bound, x_adv.detach())


# ---------------------------------------------------------------------------------------------------------------------
class DeepFool(UntargetedL2Attack):
    batched = True

    def __init__(self, num_classes=10, overshoot=0.02, max_iter=50):
        self.num_classes, self.overshoot, self.max_iter = num_classes, overshoot, max_iter

    def __call__(self, image, gt_label, net):
        B = image.shape[0]
        ar = torch.arange(B, device=image.device)
        gt = gt_label.view(-1)
        with torch.no_grad():
            f0 = net(image)
        ranked = torch.argsort(f0, dim=1, descending=True)[:, :self.num_classes]          # (B, K): each image its own top classes
        label = ranked[:, 0]
        already = gt != label                                                              # misclassified at the start: nothing to attack
        active = ~already
        r_tot = torch.zeros_like(image, dtype=torch.float32)
        pert_image = image.clone()
        k_i, it = label.clone(), 0
        while bool(active.any()) and it < self.max_iter:
            # one forward per iteration (decision AND gradients); the backward passes (one per class rank for the whole batch)
            # only once some image is known to be still active: the terminating iteration pays for the forward alone
            jac = ClassJacobian(net, pert_image, ranked)
            fs = jac.logits                                                                # (B, n)
            if it > 0:
                k_i = torch.where(active, fs.argmax(dim=1), k_i)
                active = active & (k_i == label)
                if not bool(active.any()):
                    break
            grads = jac.grads()                                                            # (B, K, C, H, W)
            w_k = grads[:, 1:] - grads[:, 0:1]                                             # (B, K-1, C, H, W) float32
            f_k = (fs.gather(1, ranked[:, 1:]) - fs.gather(1, ranked[:, :1])).abs()
            dist = f_k / w_k.flatten(2).norm(dim=2)
            j = torch.argmin(dist, dim=1)                                                  # first minimum, like the reference's strict '<' scan
            w = w_k[ar, j]
            r_i = _bc(dist[ar, j] + 1e-4, w) * w / _bc(w.flatten(1).norm(dim=1), w)
            r_new = (r_tot.double() + r_i.double()).float()                                # float64 accumulate, float32 keep (:549-550)
            r_tot = torch.where(_bc(active, r_tot), r_new, r_tot)
            pert_image = torch.where(_bc(active, r_tot), image + (1 + self.overshoot) * r_tot, pert_image)
            it += 1
        if bool(active.any()):                                                             # the decision after the last step
            with torch.no_grad():
                k_i = torch.where(active, net(pert_image).argmax(dim=1), k_i)
        fooled = (k_i != gt) & ~already
        r_fin = ((1 + self.overshoot) * r_tot).flatten(1).norm(dim=1)
        bound = torch.where(already, torch.zeros_like(r_fin), torch.where(fooled, r_fin, torch.full_like(r_fin, float('inf'))))
        adv = torch.where(_bc(fooled, image), pert_image, image).detach()
        return _ret(fooled | already, bound, adv)


# ---------------------------------------------------------------------------------------------------------------------
class CW(UntargetedL2Attack):
    batched = True

    def __init__(self, c: float = 1., kappa: float = 0., steps: int = 64, lr: float = 1e-2, n_restarts: int = 1,
                 early_stopping_steps: int = 16):
        self.c, self.kappa, self.steps, self.lr, self.n_restarts = c, kappa, steps, lr, n_restarts
        self.early_stopping_len = early_stopping_steps

    def margin(self, logits, label):
        """max(Z_y - max_{j != y} Z_j + kappa, 0) — the paper's f function."""
        one_hot = nn.functional.one_hot(label, logits.shape[1])
        real = torch.sum(one_hot * logits, 1)
        other, _ = torch.max((1 - one_hot) * logits - one_hot * 1e4, 1)
        return torch.max((real - other) + self.kappa, torch.zeros_like(real))

    def __call__(self, image, gt_label, net):
        """per image: restarts with an adaptive c, FGSM + noise start, Adam on w = atanh(2x - 1) with the gradient clipped to norm 1,
        early stop once a fooling iterate stops improving (:363-467).  Every piece of state is per image; an image that stopped
        early keeps its iterate frozen while the others go on."""
        image, label = image.clone().detach(), gt_label.clone().detach().view(-1)
        B, dev = image.shape[0], image.device
        best_ok = torch.zeros(B, dtype=torch.bool, device=dev)
        best_adv, best_l2 = image.clone(), torch.zeros(B, device=dev)
        c = torch.full((B,), float(self.c), device=dev)
        res = np.log2(image.shape[-1])
        init = FGSM(l2_bound=np.power(2, res - 5))                  # FGSM start scaled with the image size (:363-365)
        for _ in range(self.n_restarts):
            run_adv = init(image, label, net)[2]
            noise = _per_image_randn(image)
            noise = noise * np.power(2, res - 8) / _bc(torch.norm(noise.view(B, -1), dim=1), noise)
            run_adv = torch.clamp(run_adv + noise, min=1e-6, max=1 - 1e-6)
            run_l2 = (run_adv - image).flatten(1).norm(dim=1)
            w = torch.atanh(run_adv * 2. - 1).requires_grad_(True)
            opt = torch.optim.Adam([w], lr=self.lr)
            mean_loss = torch.zeros(B, device=dev)
            n_mean = torch.zeros(B, dtype=torch.long, device=dev)
            run_ok = torch.zeros(B, dtype=torch.bool, device=dev)
            first = torch.ones(B, dtype=torch.bool, device=dev)      # no iterate recorded yet ("not run_ok" of the first pass)
            active = torch.ones(B, dtype=torch.bool, device=dev)
            for _step in range(self.steps):
                with torch.enable_grad():
                    cur = 0.5 * (torch.tanh(w) + 1)
                    logits = net(cur)
                    loss = ((cur - image) ** 2).flatten(1).sum(dim=1) + c * self.margin(logits, label)      # (B,)
                    opt.zero_grad()
                    (gw,) = torch.autograd.grad(loss.sum(), [w])
                coef = torch.clamp(1.0 / (gw.flatten(1).norm(dim=1) + 1e-6), max=1.0)                      # clip_grad_norm_(max_norm=1), per image
                w.grad = gw * _bc(coef, gw)
                w_prev = w.detach().clone()
                opt.step()
                with torch.no_grad():
                    w.copy_(torch.where(_bc(active, w), w, w_prev))                                          # stopped images stay put
                fooled = torch.argmax(logits.detach(), 1) != label
                lv = loss.detach()
                stop = active & fooled & (lv > mean_loss) & (n_mean > self.early_stopping_len)               # fooling but not converging any more
                upd = active & fooled & ~stop
                look = torch.minimum(n_mean, torch.full_like(n_mean, self.early_stopping_len)).to(lv.dtype)
                mean_loss = torch.where(upd, (mean_loss * look + lv) / (look + 1), mean_loss)
                n_mean = n_mean + upd.long()
                active = active & ~stop
                this_l2 = (cur.detach() - image).flatten(1).norm(dim=1)
                take = active & (first | ~run_ok | (run_l2 > this_l2))
                run_adv = torch.where(_bc(take, run_adv), cur.detach(), run_adv)
                run_l2 = torch.where(take, this_l2, run_l2)
                run_ok = torch.where(take, fooled, run_ok)
                first = first & ~take
                if not bool(active.any()):
                    break
            with torch.no_grad():
                fooled = torch.argmax(net(run_adv), 1) != label
            improve = fooled & (~best_ok | (best_l2 > run_l2))
            worse = fooled & ~improve & (best_l2 < run_l2)
            c = torch.where(~fooled, 1.2 * c, torch.where(improve, 0.8 * c, torch.where(worse, 0.9 * c, c)))
            best_adv = torch.where(_bc(improve, best_adv), run_adv, best_adv)
            best_l2 = torch.where(improve, run_l2, best_l2)
            best_ok = best_ok | improve
            c = c.clamp(0.1, 1000.0)
        return _ret(best_ok, best_l2, best_adv)


# ---------------------------------------------------------------------------------------------------------------------
class APGDAttack(UntargetedL2Attack):
    batched = True

    def __init__(self, n_iter: int, rho: float, max_bound: float, ce_loss: bool):
        self.n_iter, self.rho, self.max_bound = n_iter, rho, max_bound
        self.criterion = nn.CrossEntropyLoss(reduction='none') if ce_loss else self.dlr_loss
        self.division_eps = 1e-12
        self.initial_step_size_iters = max(int(0.22 * n_iter), 1)
        self.min_step_size_iters = max(int(0.06 * n_iter), 1)
        self.step_size_decr = max(int(0.03 * n_iter), 1)

    def dlr_loss(self, logits, gt_label):
        """-(z_y - max_{j != y} z_j) / (z_(1) - z_(3)), with the reference's guard when z_(3) is the true class (:86-123);
        per image"""
        if logits.shape[1] < 4:
            raise AttributeError('APGD_DLR is undefined for problems with less than 4 classes!')
        gt = gt_label.view(-1)
        srt, idx = logits.sort(dim=1)
        still_correct = torch.eq(idx[:, -1], gt)
        z_y = logits.gather(1, gt.view(-1, 1)).squeeze(1)
        z_other = torch.where(still_correct, srt[:, -2], srt[:, -1])
        third = torch.where(torch.ne(srt[:, -3], z_y), srt[:, -3], srt[:, -4])
        return -(z_y - z_other) / (srt[:, -1] - third + self.division_eps)

    def _loss_and_grad(self, net, x, label):
        x = x.detach().requires_grad_(True)
        with torch.enable_grad():
            loss = self.criterion(net(x), label.view(-1))                   # (B,): rows are independent, so the gradient of the
            (g,) = torch.autograd.grad(loss.sum(), [x])                    # sum holds every image's own gradient
        return x, loss.detach(), g.detach()

    def _project(self, delta):
        """onto the L2 ball of radius max_bound (per sample)"""
        return normalize(delta) * torch.min(self.max_bound * torch.ones_like(delta), l2_norm(delta, keepdim=True))

    def _stalled(self, losses, step, lookback):
        window = losses[step - (lookback - 1): step + 1]                    # (lookback, B)
        prev = torch.roll(window, shifts=1, dims=0)
        prev[0] = window[0]
        return torch.gt(window, prev).sum(dim=0) < lookback * self.rho

    def __call__(self, image, gt_label, net, init_noise: torch.Tensor = None):
        """init_noise: the N(0,1) draw of the random start (B,3,H,W); None draws it image by image"""
        B = image.shape[0]
        noise = _per_image_randn(image) if init_noise is None else init_noise
        x_adv = (image + self.max_bound * normalize(noise)).clamp(0., 1.)
        x_prev = x_adv.clone()
        x_adv, loss, grad = self._loss_and_grad(net, x_adv, gt_label)
        step = torch.full((B,), 2 * self.max_bound, device=image.device, dtype=image.dtype)
        since_check, check_every = 0, self.initial_step_size_iters
        losses = torch.zeros([self.n_iter, B], device=image.device)
        reduced_last = torch.ones(B, dtype=torch.bool, device=image.device)
        best_loss, prev_best = loss.clone(), loss.clone()
        x_best, g_best = x_adv.clone(), grad.clone()
        for i in range(self.n_iter):
            with torch.no_grad():
                x_adv = x_adv.detach()
                momentum = x_adv - x_prev
                x_prev = x_adv.clone()
                a = 0.75 if i > 0 else 1.0
                z = torch.clamp(image + self._project(x_adv + _bc(step, image) * normalize(grad) - image), 0., 1.)
                z = x_adv + (z - x_adv) * a + momentum * (1 - a)
                x_adv = torch.clamp(image + self._project(z - image), 0., 1.)
            x_adv, loss, grad = self._loss_and_grad(net, x_adv, gt_label)
            with torch.no_grad():
                losses[i] = loss
                better = loss > best_loss
                best_loss = torch.where(better, loss, best_loss)
                x_best = torch.where(_bc(better, x_adv), x_adv, x_best)
                g_best = torch.where(_bc(better, grad), grad, g_best)
                since_check += 1
                if since_check == check_every:
                    halve = self._stalled(losses, i, since_check) | ((prev_best >= best_loss) & ~reduced_last)
                    reduced_last, prev_best = halve, best_loss.clone()
                    step = torch.where(halve, step / 2.0, step)
                    x_adv = torch.where(_bc(halve, x_adv), x_best, x_adv)
                    grad = torch.where(_bc(halve, grad), g_best, grad)
                    since_check = 0
                    check_every = max(check_every - self.step_size_decr, self.min_step_size_iters)
        with torch.no_grad():
            ok = torch.ne(net(x_adv).argmax(dim=1), gt_label.view(-1))
        bound = (x_adv.detach() - image.detach()).flatten(1).norm(dim=1)
        return _ret(ok, bound, x_adv.detach())


# ---------------------------------------------------------------------------------------------------------------------
class FABAttack(UntargetedL2Attack):
    batched = True

    def __init__(self, n_iter: int, alpha_max: float, eta: float, beta: float):
        self.n_iter, self.eta, self.beta, self.alpha_max = n_iter, eta, beta, alpha_max

    def get_diff_logits_grads(self, image, label, net):
        """logit differences to the true class and their input-gradients, all classes from ONE forward of the batch and one
        backward pass per class for all images (:605-635 loops over the classes per image)"""
        y, g = class_gradients(net, image)                          # (B, n), (B, n, C, H, W)
        ar = torch.arange(y.shape[0], device=y.device)
        df = y - y[ar, label].unsqueeze(1)
        dg = g - g[ar, label].unsqueeze(1)
        df[ar, label] = 1e10
        return df, dg

    def __call__(self, image, gt_label, net):
        image = image.detach().clone()
        B = image.shape[0]
        ar = torch.arange(B, device=image.device)
        gt = gt_label.view(-1)
        with torch.no_grad():
            already = net(image).argmax(dim=1) != gt
        if bool(already.all()):
            return _ret(already, torch.zeros(B, device=image.device), image.detach())
        x_adv = image.clone()
        bound = torch.full((B,), 1e10, device=image.device)
        ok = torch.zeros(B, dtype=torch.bool, device=image.device)
        x_orig, x_i = image.clone(), image.clone()
        flat_orig = image.clone().view(B, -1)
        for _ in range(self.n_iter):
            df, dg = self.get_diff_logits_grads(x_i, gt, net)
            with torch.no_grad():
                dist = df.abs() / (1e-12 + (dg ** 2).reshape(B, df.shape[1], -1).sum(dim=-1).sqrt())
                s = dist.min(dim=1).indices                                   # closest decision hyperplane, per image
                dg_s = dg[ar, s]
                b = -df[ar, s] + (dg_s * x_i).view(B, -1).sum(dim=-1)
                w = dg_s.view(B, -1)
                d3 = projection_l2(torch.cat((x_i.view(B, -1), flat_orig), 0), torch.cat((w, w), 0), torch.cat((b, b), 0).unsqueeze(1))
                d_i, d_o = torch.reshape(d3[:B], x_i.shape), torch.reshape(d3[B:], x_i.shape)
                a0 = (d3 ** 2).sum(dim=1, keepdim=True).sqrt().view(-1, 1, 1, 1)
                a0 = torch.max(a0, 1e-8 * torch.ones_like(a0))
                a1, a2 = a0[:B], a0[B:]
                alpha = torch.min(torch.max(a1 / (a1 + a2), torch.zeros_like(a1)), self.alpha_max * torch.ones_like(a1))
                x_i = ((x_i + self.eta * d_i) * (1 - alpha) + (x_orig + d_o * self.eta) * alpha).clamp(0.0, 1.0)
                fooled = torch.ne(net(x_i).argmax(dim=1), gt) & ~already
                ok = ok | fooled
                t = ((x_i - x_orig) ** 2).view(B, -1).sum(dim=-1).sqrt()
                improve = fooled & (t < bound)
                x_adv = torch.where(_bc(improve, x_adv), x_i, x_adv)
                bound = torch.where(improve, t, bound)
                x_i = torch.where(_bc(fooled, x_i), (1 - self.beta) * x_orig + self.beta * x_i, x_i)   # step back towards the original
        ok = ok | already
        bound = torch.where(already, torch.zeros_like(bound), bound)
        return _ret(ok, bound, x_adv.detach())


# ---------------------------------------------------------------------------------------------------------------------
class AutoAttack(UntargetedL2Attack):
    """APGD-CE at bounds .5/1/4, APGD-DLR at .5/2/4 (when > 3 classes), FAB; smallest successful distortion wins.  Batched:
    an image leaves an escalation as soon as one bound succeeds for it, the remaining images go on as a smaller batch."""
    batched = True

    def __init__(self):
        mk = lambda bound, ce: APGDAttack(n_iter=64, rho=0.75, max_bound=bound, ce_loss=ce)   # noqa: E731
        self.ce = [mk(0.5, True), mk(1.0, True), mk(4.0, True)]
        self.dlr = [mk(0.5, False), mk(2.0, False), mk(4.0, False)]
        self.fab = FABAttack(n_iter=128, alpha_max=0.1, eta=1.05, beta=0.9)

    @staticmethod
    def _tensors(res, image):
        s, b, a = res
        if not torch.is_tensor(s):
            s = torch.tensor([bool(s)], device=image.device)
            b = torch.tensor([float(b)], device=image.device)
        return s.clone(), b.clone().float(), a.clone()

    @staticmethod
    def _better(cur, new):
        """per image: a success replaces a failure (:267-285); between two successes the smaller distortion wins"""
        (s0, b0, a0), (s1, b1, a1) = cur, new
        first = s1 & ~s0
        smaller = s1 & s0 & (b1 < b0)
        take = first | smaller
        return s0 | s1, torch.where(take, b1, b0), torch.where(_bc(take, a0), a1, a0)

    def _escalate(self, attacks: List[APGDAttack], image, label, net):
        """try increasing bounds until one succeeds (:288-296, :307-316), image by image"""
        res = self._tensors(attacks[0](image, label, net), image)
        for atk in attacks[1:]:
            todo = (~res[0]).nonzero().flatten()
            if todo.numel() == 0:
                break
            sub = self._tensors(atk(image[todo], label.view(-1)[todo], net), image)
            s, b, a = res
            bs, bb, ba = self._better((s[todo], b[todo], a[todo]), sub)
            s[todo], b[todo], a[todo] = bs, bb, ba
            res = (s, b, a)
        return res

    def __call__(self, image, gt_label, net):
        best = self._escalate(self.ce, image, gt_label, net)
        with torch.no_grad():
            n_classes = net(image[:1]).shape[1]
        if n_classes > 3:
            best = self._better(best, self._escalate(self.dlr, image, gt_label, net))
        best = self._better(best, self._tensors(self.fab(image, gt_label, net), image))
        return _ret(*best)
