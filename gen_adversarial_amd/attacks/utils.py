"""
Helpers of the L2 attacks (reference: src/attacks/utils.py:6-76).  The reference's helpers act on ONE image (whole-tensor
norms); the versions here are per-sample over dim 0 so that a batch of B images is B independent attacks, and reduce to the
reference exactly for B = 1.
"""
from __future__ import annotations

import math

import torch


def l2_norm(x: torch.Tensor, keepdim: bool = False) -> torch.Tensor:
    """per-sample L2 norm; src/attacks/utils.py:6-11 for a batch of one."""
    z = (x ** 2).flatten(1).sum(dim=1).sqrt()
    return z.view(-1, *([1] * (x.dim() - 1))) if keepdim else z


def normalize(x: torch.Tensor) -> torch.Tensor:
    """x / ||x||_2 per sample; src/attacks/utils.py:14-19."""
    return x / l2_norm(x, keepdim=True)


def projection_l2(points: torch.Tensor, w_hyperplane: torch.Tensor, b_hyperplane: torch.Tensor) -> torch.Tensor:
    """
    For every row: the smallest-L2 step d with  <w, p + d> = b  and  0 <= p + d <= 1  (FAB's box-constrained projection on a
    hyperplane; Croce & Hein 2020, restating src/attacks/utils.py:22-76).  Rows are independent.

    Sketch: orient w so that the current value c = <w,p> - b is >= 0; moving coordinate i by the signed amount that hits
    its box face costs r_i per unit of w_i; sort those ratios, accumulate the attainable change of <w, .> (piecewise linear in
    the Lagrange multiplier) and locate by bisection the segment in which the constraint is met.
    """
    t, w, b = points, w_hyperplane.clone(), b_hyperplane
    n = w.shape[1]

    c = (w * t).sum(dim=1) - b[:, 0]
    sign = 2 * (c >= 0) - 1
    w.mul_(sign.unsqueeze(1))
    c.mul_(sign)

    small = w.abs() < 1e-8
    r = torch.max(t / w, (t - 1) / w).clamp(min=-1e12, max=1e12)
    r.masked_fill_(small, 1e12)
    r[r == -1e12] *= -1
    rs, order = torch.sort(r, dim=1)
    rs_next = torch.nn.functional.pad(rs[:, 1:], (0, 1))
    rs.masked_fill_(rs == 1e12, 0)
    rs_next.masked_fill_(rs_next == 1e12, 0)

    w2_sorted = (w ** 2).gather(1, order)
    w2_total = w2_sorted.sum(dim=1, keepdim=True)
    w2_rest = w2_total - torch.cumsum(w2_sorted, dim=1)
    d = -(r * w)
    d.mul_((~small).float())
    s = torch.cat((-w2_total * rs[:, 0:1], torch.cumsum((-rs_next + rs) * w2_rest, dim=1) - w2_total * rs[:, 0:1]), 1)

    below = s[:, 0] + c < 0                     # the unconstrained projection already satisfies the box
    above = (d * w).sum(dim=1) + c > 0          # even the full box move cannot reach the hyperplane
    mid = ~(below | above)

    lo = torch.zeros(int(mid.sum()), device=t.device)
    hi = torch.full_like(lo, n - 1)
    s_mid, c_mid = s[mid], c[mid]
    for _ in range(math.ceil(math.log2(n))):
        probe = torch.floor((lo + hi) / 2)
        cond = s_mid.gather(1, probe.long().unsqueeze(1)).squeeze(1) + c_mid > 0
        lo = torch.where(cond, probe, lo)
        hi = torch.where(cond, hi, probe)
    lo = lo.long()

    if below.any():
        alpha = c[below] / w2_total[below].squeeze(-1)
        d[below] = -alpha.unsqueeze(-1) * w[below]
    if mid.any():
        alpha = (s[mid, lo] + c[mid]) / w2_rest[mid, lo] + rs[mid, lo]
        alpha[w2_rest[mid, lo] == 0] = 0
        keep = (alpha.unsqueeze(-1) > r[mid]).float()
        d[mid] = d[mid] * keep - alpha.unsqueeze(-1) * w[mid] * (1 - keep)
    return d * (~small).float()
