"""
PGD-Linf under the reference's attack protocol (`attack(image, gt_label, net) -> (success, bound, adv)`,
src/attacks/untargeted.py:13-34).  The reference tree has no PGD-Linf evaluation attack (its only
L-inf PGD is the TRADES inner loop, src/defenses/competitors/trades/modules.py:36-45); this is new code written to the
same protocol and to the setting BASELINE.json names (eps = 8/255).  It is batched over images: `image` may be
(B,3,H,W) with `gt_label` (B,), every image attacked independently (per-image early stop masks), so that a GPU is
kept busy with B x EoT defender rows per step; B = 1 reproduces the reference's one-image-at-a-time loop.
"""
from __future__ import annotations

from contextlib import contextmanager

import torch


@contextmanager
def bpda(net: torch.nn.Module):
    """Backward Pass Differentiable Approximation (Athalye, Carlini & Wagner 2018) for the attack BASELINE.json names
    ("PGD-40 + BPDA"; the reference tree has neither): inside the block, gradients through the MLVGM defender
    wrapped by `net` (directly or inside an EoTWrapper) count the purifier as the identity — forward passes stay exact."""
    target = getattr(net, 'model', net)
    if not hasattr(target, 'bpda'):
        raise TypeError('BPDA needs an MLVGM defender (a purifier in front of a classifier); '
                        f'{type(target).__name__} has none')
    old = target.bpda
    target.bpda = True
    try:
        yield net
    finally:
        target.bpda = old


class PGDLinf:
    batched = True      # `image` may hold several images: each is attacked independently (see test_defense.evaluate_shard)

    def __init__(self, eps: float = 8.0 / 255.0, step_size: float = 2.0 / 255.0, steps: int = 40,
                 random_start: bool = False, bpda: bool = False):
        self.eps, self.step_size, self.steps, self.random_start = eps, step_size, steps, random_start
        self.bpda = bpda            # gradients with the purifier counted as the identity (see bpda() above)

    @staticmethod
    def step(x_adv: torch.Tensor, x_orig: torch.Tensor, grad: torch.Tensor, eps: float, step_size: float,
             active: torch.Tensor = None) -> torch.Tensor:
        """one ascent step on the loss + projection on the eps-ball and on [0,1]"""
        nxt = x_adv + step_size * grad.sign()
        nxt = torch.min(torch.max(nxt, x_orig - eps), x_orig + eps).clamp(0.0, 1.0)
        if active is not None:
            nxt = torch.where(active.view(-1, 1, 1, 1), nxt, x_adv)
        return nxt

    def __call__(self, image: torch.Tensor, gt_label: torch.Tensor, net: torch.nn.Module):
        if self.bpda:
            with bpda(net):
                return self._run(image, gt_label, net)
        return self._run(image, gt_label, net)

    def _run(self, image: torch.Tensor, gt_label: torch.Tensor, net: torch.nn.Module):
        x_orig = image.detach()
        x_adv = x_orig.clone()
        if self.random_start:
            x_adv = (x_adv + torch.empty_like(x_adv).uniform_(-self.eps, self.eps)).clamp(0, 1)
        B = x_orig.shape[0]
        success = torch.zeros(B, dtype=torch.bool, device=x_orig.device)
        best = x_orig.clone()
        for _ in range(self.steps):
            x_adv.requires_grad_(True)
            logits = net(x_adv)
            wrong = logits.argmax(dim=-1) != gt_label
            newly = wrong & ~success
            best = torch.where(newly.view(-1, 1, 1, 1), x_adv.detach(), best)
            success |= wrong
            if bool(success.all()):
                break
            loss = torch.nn.functional.cross_entropy(logits, gt_label, reduction='sum')
            (grad,) = torch.autograd.grad(loss, [x_adv])
            x_adv = self.step(x_adv.detach(), x_orig, grad, self.eps, self.step_size, ~success)
        else:
            with torch.no_grad():
                wrong = net(x_adv) .argmax(dim=-1) != gt_label
            best = torch.where((wrong & ~success).view(-1, 1, 1, 1), x_adv.detach(), best)
            success |= wrong
        bound = (best - x_orig).flatten(1).abs().amax(dim=1)
        if B == 1:
            return bool(success.item()), float(bound.item()), best
        return success, bound, best
