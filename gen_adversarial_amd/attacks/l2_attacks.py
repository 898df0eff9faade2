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
Random draws use `torch.randn_like(image)` in the reference's order, so a CPU run under `torch.manual_seed` reproduces
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


def class_gradients(net: nn.Module, x: torch.Tensor, classes=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """ONE forward, then d logits[0, k] / d x for every k in `classes` (all classes when None): logits (1, n) and the
    gradients stacked (len(classes), *x.shape[1:]).  With a stochastic defender every gradient belongs to the same draw."""
    x = x.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        y = net(x)
        ks = range(y.shape[1]) if classes is None else classes
        grads = [torch.autograd.grad(y[0, int(k)], [x], retain_graph=True)[0][0] for k in ks]
    return y.detach(), torch.stack(grads, dim=0)


# ---------------------------------------------------------------------------------------------------------------------
class FGSM(UntargetedL2Attack):
    def __init__(self, l2_bound: float):
        self.l2_bound = l2_bound

    def __call__(self, image, gt_label, net):
        x = image.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            logits = net(x)
            if torch.argmax(logits, dim=-1) != gt_label:
                return True, 0.0, image
            cost = -nn.functional.cross_entropy(logits, gt_label)
            (g,) = torch.autograd.grad(cost, [x])
        step = g.sign()
        step = step / torch.norm(step.view(step.size(0), -1), p=2, dim=1, keepdim=True).view(-1, 1, 1, 1)
        x_adv = torch.clamp(x.detach() - step * self.l2_bound, 0., 1.)
        with torch.no_grad():
            fooled = torch.argmax(net(x_adv), dim=-1) != gt_label
        return fooled, self.l2_bound, x_adv.detach()


# ---------------------------------------------------------------------------------------------------------------------
class DeepFool(UntargetedL2Attack):
    def __init__(self, num_classes=10, overshoot=0.02, max_iter=50):
        self.num_classes, self.overshoot, self.max_iter = num_classes, overshoot, max_iter

    def __call__(self, image, gt_label, net):
        with torch.no_grad():
            f0 = net(image).flatten()
        ranked = torch.argsort(f0, descending=True)[:self.num_classes]
        label = int(ranked[0])
        if int(gt_label) != label:
            return True, 0.0, image.detach()                       # already misclassified: nothing to attack

        r_tot = torch.zeros_like(image, dtype=torch.float32)
        pert_image = image.clone()
        k_i, it = label, 0
        x = pert_image.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            fs = net(x)                                            # one forward per iteration: decision AND gradients
        while k_i == label and it < self.max_iter:
            with torch.enable_grad():
                grads = torch.stack([torch.autograd.grad(fs[0, int(k)], [x], retain_graph=True)[0][0] for k in ranked])
            w_k = grads[1:] - grads[0:1]                           # (k-1, C, H, W) float32
            f_k = (fs.detach()[0, ranked[1:]] - fs.detach()[0, ranked[0]]).abs()
            dist = f_k / w_k.flatten(1).norm(dim=1)
            j = int(torch.argmin(dist))                            # first minimum, like the reference's strict '<' scan
            w = w_k[j]
            r_i = (dist[j] + 1e-4) * w / w.flatten().norm()
            r_tot = (r_tot.double() + r_i.double().unsqueeze(0)).float()      # float64 accumulate, float32 keep (:549-550)
            pert_image = image + (1 + self.overshoot) * r_tot
            x = pert_image.detach().clone().requires_grad_(True)
            with torch.enable_grad():
                fs = net(x)
            k_i = int(torch.argmax(fs.detach().flatten()))
            it += 1
        if k_i == int(gt_label):
            return False, float('inf'), image.detach()
        r_fin = (1 + self.overshoot) * r_tot
        return True, float(r_fin.flatten().norm()), pert_image.detach()


# ---------------------------------------------------------------------------------------------------------------------
class CW(UntargetedL2Attack):
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
        image, label = image.clone().detach(), gt_label.clone().detach()
        best_ok, best_adv, best_l2 = False, image.clone(), 0.
        c = self.c
        res = np.log2(image.shape[-1])
        init = FGSM(l2_bound=np.power(2, res - 5))                  # FGSM start scaled with the image size (:363-365)
        for _ in range(self.n_restarts):
            run_adv = init(image, label, net)[2]
            noise = torch.randn_like(image)
            noise = noise * np.power(2, res - 8) / torch.norm(noise.view(1, -1), dim=1, keepdim=True)
            run_adv = torch.clamp(run_adv + noise, min=1e-6, max=1 - 1e-6)
            run_l2 = torch.linalg.norm((run_adv - image).flatten(), ord=2)
            w = torch.atanh(run_adv * 2. - 1).requires_grad_(True)
            opt = torch.optim.Adam([w], lr=self.lr)
            mean_loss, n_mean, run_ok = 0.0, 0, False
            for _step in range(self.steps):
                with torch.enable_grad():
                    cur = 0.5 * (torch.tanh(w) + 1)
                    logits = net(cur)
                    loss = nn.functional.mse_loss(cur, image, reduction='sum') + c * self.margin(logits, label)
                    opt.zero_grad()
                    (gw,) = torch.autograd.grad(loss, [w])
                w.grad = gw
                torch.nn.utils.clip_grad_norm_([w], max_norm=1.)
                opt.step()
                fooled = bool((torch.argmax(logits.detach(), 1) != label).item())
                if fooled:
                    lv = loss.detach().item()
                    if lv > mean_loss and n_mean > self.early_stopping_len:
                        break                                       # fooling but not converging any more
                    look = min(n_mean, self.early_stopping_len)
                    mean_loss = (mean_loss * look + lv) / (look + 1)
                    n_mean += 1
                this_l2 = torch.linalg.norm((cur.detach() - image).flatten(), ord=2)
                if not run_ok or run_l2 > this_l2:
                    run_adv, run_l2, run_ok = cur.detach(), this_l2, fooled
            with torch.no_grad():
                fooled = bool((torch.argmax(net(run_adv), 1) != label).item())
            if not fooled:
                c = 1.2 * c
            elif (not best_ok) or best_l2 > run_l2:
                c = 0.8 * c
                best_adv, best_l2, best_ok = run_adv, run_l2, True
            elif best_l2 < run_l2:
                c = 0.9 * c
            c = max(min(c, 1000), 0.1)
        return best_ok, best_l2, best_adv


# ---------------------------------------------------------------------------------------------------------------------
class APGDAttack(UntargetedL2Attack):
    def __init__(self, n_iter: int, rho: float, max_bound: float, ce_loss: bool):
        self.n_iter, self.rho, self.max_bound = n_iter, rho, max_bound
        self.criterion = nn.CrossEntropyLoss(reduction='none') if ce_loss else self.dlr_loss
        self.division_eps = 1e-12
        self.initial_step_size_iters = max(int(0.22 * n_iter), 1)
        self.min_step_size_iters = max(int(0.06 * n_iter), 1)
        self.step_size_decr = max(int(0.03 * n_iter), 1)

    def dlr_loss(self, logits, gt_label):
        """-(z_y - max_{j != y} z_j) / (z_(1) - z_(3)), with the reference's guard when z_(3) is the true class (:86-123)."""
        if logits.shape[1] < 4:
            raise AttributeError('APGD_DLR is undefined for problems with less than 4 classes!')
        srt, idx = logits.sort(dim=1)
        still_correct = bool(torch.eq(idx[:, -1], gt_label).item())
        z_y = logits[0, gt_label]
        z_other = srt[:, -2] if still_correct else srt[:, -1]
        third = srt[:, -3] if bool(torch.ne(srt[:, -3], z_y)) else srt[:, -4]
        return -(z_y - z_other) / (srt[:, -1] - third + self.division_eps)

    def _loss_and_grad(self, net, x, label):
        x = x.detach().requires_grad_(True)
        with torch.enable_grad():
            loss = self.criterion(net(x), label)
            (g,) = torch.autograd.grad(loss, [x])
        return x, loss, g.detach()

    def _project(self, delta):
        """onto the L2 ball of radius max_bound (per sample)"""
        return normalize(delta) * torch.min(self.max_bound * torch.ones_like(delta), l2_norm(delta, keepdim=True))

    def _stalled(self, losses, step, lookback):
        window = losses[step - (lookback - 1): step + 1]
        prev = torch.roll(window, shifts=1, dims=0)
        prev[0] = window[0]
        return torch.gt(window, prev).sum().item() < lookback * self.rho

    def __call__(self, image, gt_label, net):
        x_adv = (image + self.max_bound * normalize(torch.randn_like(image))).clamp(0., 1.)
        x_prev = x_adv.clone()
        x_adv, loss, grad = self._loss_and_grad(net, x_adv, gt_label)
        step = 2 * self.max_bound
        since_check, check_every = 0, self.initial_step_size_iters
        losses = torch.zeros([self.n_iter, 1], device=image.device)
        reduced_last = True
        best_loss = prev_best = loss.item()
        x_best, g_best = x_adv.clone(), grad.clone()
        for i in range(self.n_iter):
            with torch.no_grad():
                x_adv = x_adv.detach()
                momentum = x_adv - x_prev
                x_prev = x_adv.clone()
                a = 0.75 if i > 0 else 1.0
                z = torch.clamp(image + self._project(x_adv + step * normalize(grad) - image), 0., 1.)
                z = x_adv + (z - x_adv) * a + momentum * (1 - a)
                x_adv = torch.clamp(image + self._project(z - image), 0., 1.)
            x_adv, loss, grad = self._loss_and_grad(net, x_adv, gt_label)
            with torch.no_grad():
                lv = loss.item()
                losses[i] = lv
                if lv > best_loss:
                    best_loss, x_best, g_best = lv, x_adv.clone(), grad.clone()
                since_check += 1
                if since_check == check_every:
                    halve = self._stalled(losses, i, since_check) or (prev_best >= best_loss and not reduced_last)
                    reduced_last, prev_best = halve, best_loss
                    if halve:
                        step /= 2.0
                        x_adv, grad = x_best.clone(), g_best.clone()
                    since_check = 0
                    check_every = max(check_every - self.step_size_decr, self.min_step_size_iters)
        with torch.no_grad():
            ok = torch.ne(net(x_adv).argmax(dim=1), gt_label).item()
        bound = torch.linalg.norm((x_adv.detach() - image.detach()).flatten(), ord=2).item()
        return ok, bound, x_adv.detach()


# ---------------------------------------------------------------------------------------------------------------------
class FABAttack(UntargetedL2Attack):
    def __init__(self, n_iter: int, alpha_max: float, eta: float, beta: float):
        self.n_iter, self.eta, self.beta, self.alpha_max = n_iter, eta, beta, alpha_max

    def get_diff_logits_grads(self, image, label, net):
        """logit differences to the true class and their input-gradients, all classes from one forward (:605-635)."""
        y, g = class_gradients(net, image)                          # g: (n_classes, C, H, W)
        g = g.unsqueeze(0)                                           # (1, n_classes, C, H, W)
        df = y - y[:, label]
        dg = g - g[:, label]
        df[:, label] = 1e10
        return df, dg

    def __call__(self, image, gt_label, net):
        image = image.detach().clone()
        with torch.no_grad():
            if torch.argmax(net(image)) != gt_label:
                return True, 0.0, image.detach()
        x_adv, bound, ok = image.clone(), 1e10, False
        x_orig, x_i = image.clone(), image.clone()
        flat_orig = image.clone().view(1, -1)
        for _ in range(self.n_iter):
            df, dg = self.get_diff_logits_grads(x_i, gt_label, net)
            with torch.no_grad():
                dist = df.abs() / (1e-12 + (dg ** 2).reshape(1, df.shape[1], -1).sum(dim=-1).sqrt())
                s = dist.min(dim=1).indices                                   # closest decision hyperplane
                dg_s = dg[:, s]
                b = -df[:, s] + (dg_s * x_i).view(1, -1).sum(dim=-1)
                w = dg_s.view([1, -1])
                d3 = projection_l2(torch.cat((x_i.view(1, -1), flat_orig), 0), torch.cat((w, w), 0), torch.cat((b, b), 0))
                d_i, d_o = torch.reshape(d3[:1], x_i.shape), torch.reshape(d3[-1:], x_i.shape)
                a0 = (d3 ** 2).sum(dim=1, keepdim=True).sqrt().view(-1, 1, 1, 1)
                a0 = torch.max(a0, 1e-8 * torch.ones_like(a0))
                a1, a2 = a0[:1], a0[-1:]
                alpha = torch.min(torch.max(a1 / (a1 + a2), torch.zeros_like(a1)), self.alpha_max * torch.ones_like(a1))
                x_i = ((x_i + self.eta * d_i) * (1 - alpha) + (x_orig + d_o * self.eta) * alpha).clamp(0.0, 1.0)
                if torch.ne(net(x_i).argmax(dim=1), gt_label).item():
                    ok = True
                    t = ((x_i - x_orig) ** 2).view(1, -1).sum(dim=-1).sqrt().item()
                    if t < bound:
                        x_adv, bound = x_i.clone(), t
                    x_i = (1 - self.beta) * x_orig + self.beta * x_i          # step back towards the original
        return ok, bound, x_adv.detach()


# ---------------------------------------------------------------------------------------------------------------------
class AutoAttack(UntargetedL2Attack):
    """APGD-CE at bounds .5/1/4, APGD-DLR at .5/2/4 (when > 3 classes), FAB; smallest successful distortion wins."""

    def __init__(self):
        mk = lambda bound, ce: APGDAttack(n_iter=64, rho=0.75, max_bound=bound, ce_loss=ce)   # noqa: E731
        self.ce = [mk(0.5, True), mk(1.0, True), mk(4.0, True)]
        self.dlr = [mk(0.5, False), mk(2.0, False), mk(4.0, False)]
        self.fab = FABAttack(n_iter=128, alpha_max=0.1, eta=1.05, beta=0.9)

    @staticmethod
    def _better(cur, new):
        (s0, b0, a0), (s1, b1, a1) = cur, new
        if s1 and not s0:
            return new
        if s1 and s0 and b1 < b0:
            return s0, b1, a1
        return cur

    def _escalate(self, attacks: List[APGDAttack], image, label, net):
        """try increasing bounds until one succeeds (:288-296, :307-316)"""
        res = attacks[0](image, label, net)
        for atk in attacks[1:]:
            if res[0]:
                break
            res = self._better(res, atk(image, label, net))
        return res

    def __call__(self, image, gt_label, net):
        best = self._escalate(self.ce, image, gt_label, net)
        with torch.no_grad():
            n_classes = net(image).shape[1]
        if n_classes > 3:
            best = self._better(best, self._escalate(self.dlr, image, gt_label, net))
        return self._better(best, self.fab(image, gt_label, net))
