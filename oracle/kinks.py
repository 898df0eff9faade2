"""
ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/nvae_oracle.py header for the rules).

The non-smooth points of the path — ReLU / LeakyReLU / PReLU at 0, clamp at its bounds, max-pool where the two largest
elements of a window meet — are where a 1-ulp difference in a FORWARD value (summation order, split-bf16 products) flips
a gradient routing decision, moving one gradient contribution by O(1) instead of O(ulp).  The reference's own result
has the same property (cuDNN / TF32 vs CPU), so exact gradient parity is only defined away from such ties.

This module makes that statement testable.  The oracle's kinked functions go through the wrappers below.  In normal
mode they ARE the torch functions (F.relu, F.leaky_relu, F.prelu, torch.clamp, F.max_pool2d).  Inside `flipped(delta)`
every decision that lies within `delta` of its tie is taken the OTHER way (forward values unchanged up to delta):
    tie_mask = |grad_normal - grad_flipped| > threshold
marks the gradient elements that depend on a near-tie decision.  In networks whose receptive field is the whole image most
elements do (measured: 60-95 % at delta = 10x the forward error), so the parity tests use the sharper form below and keep
`flipped` for single layers.

`replaying(candidates)` (class Replay): the oracle is evaluated with the decisions the HIP engine actually took, read from
the engine's own stored activations.  The engine's gradient must then equal the oracle's on EVERY element at the full
tolerance, and the decisions may differ from the oracle's own only within a stated distance of a tie.
"""
from __future__ import annotations

from contextlib import contextmanager

import torch
import torch.nn.functional as F

_STATE = {'delta': 0.0, 'flip': False, 'count': 0, 'total': 0, 'replay': None}


class Replay:
    """Decision replay: evaluate the oracle with the decisions ANOTHER implementation took (the HIP engine's), read from that
    implementation's own stored tensors.  `candidates` are its activation tensors as NCHW CPU tensors (pre- or post-activation:
    only signs / window arg-maxima are read).  At every kink site the oracle's input is matched, by shape and by agreement of
    the decisions themselves (>= `min_agree`), to one candidate; the candidate's decisions are then used for values AND
    gradient routing.  Sites without a match keep the oracle's own decisions.  Recorded per run:
      sites / matched      kink sites seen / matched to a candidate
      flips                decisions that differ from the oracle's own
      worst_margin         max over the flips of the oracle's distance to the tie, relative to the site tensor's max |x|
    so a test can assert (a) the other implementation's gradient equals the oracle's GIVEN its decisions, at full tolerance and
    on every element, and (b) its decisions differ from the oracle's only within `worst_margin` of a tie."""

    def __init__(self, candidates, min_agree: float = 0.999, small=()):
        self.cands = [c for c in candidates if c.dim() == 4]
        self.min_agree = min_agree
        # tensors with fewer than 48 decisions (squeeze-and-excite hidden units, [N, C/16]) cannot be identified by the agreement
        # of their decisions alone; the other implementation hands them over IN CALL ORDER (`small`: [N, F] tensors) and a site
        # takes the next one of its shape that disagrees on at most two decisions
        self.small, self.small_pos = list(small), 0
        self.sites = self.matched = self.flips = self.decisions = 0
        self.worst_margin = 0.0
        self.unmatched = []

    def _shaped(self, x):
        """candidates viewed at x's shape: more channels (padded pitch) are sliced, fewer rows (tensors shared by EoT replicas)
        are repeated"""
        n, c, h, w = x.shape
        for e in self.cands:
            if e.shape[2:] != (h, w) or e.shape[1] < c or n % e.shape[0]:
                continue
            e = e[:, :c]
            yield e if e.shape[0] == n else e.repeat_interleave(n // e.shape[0], dim=0)

    def sign(self, x: torch.Tensor, what: str):
        """decision tensor (x > 0 as the other implementation saw it) or None"""
        self.sites += 1
        if x.dim() == 2:                                        # fully connected layers: the engine keeps [N,1,1,F]
            d2 = self.sign(x[:, :, None, None], what)
            self.sites -= 1
            return None if d2 is None else d2[:, :, 0, 0]
        if x.dim() == 3:                                        # token tensors [B, T, C]: the engine keeps [B, T, 1, C] (NCHW: [B, C, T, 1])
            d3 = self.sign(x.permute(0, 2, 1).unsqueeze(-1), what)
            self.sites -= 1
            return None if d3 is None else d3.squeeze(-1).permute(0, 2, 1)
        own = x > 0
        if x.dim() != 4:
            self.unmatched.append((what, tuple(x.shape)))
            return None
        best, best_agree = None, 0.0
        if x.shape[2:] == (1, 1):
            # [N, F] tensors (squeeze-and-excite hidden units of any width): the other implementation hands them over in call
            # order; a site takes the next one of its shape that disagrees on at most two decisions.  Tried FIRST, so that the
            # position in the list follows the call order whatever the width (a wide site matched by agreement below would
            # leave it behind)
            for i in range(self.small_pos, min(self.small_pos + 4, len(self.small))):
                e = self.small[i]
                if tuple(e.shape) == tuple(x.shape[:2]) and int(((e > 0) != own[:, :, 0, 0]).sum()) <= 2:
                    best, self.small_pos = e[:, :, None, None], i + 1
                    break
        if best is None:
            for e in self._shaped(x):
                agree = ((e > 0) == own).float().mean().item()
                if agree > best_agree:
                    best, best_agree = e, agree
            # a match may disagree on at most 0.1 % of the decisions (one, for small tensors); tensors with fewer than 48 decisions
            # are never matched by agreement (a chance agreement would replay a stranger's decisions)
            ok = (best is not None and own.numel() >= 48 and
                  (1.0 - best_agree) * own.numel() <= max(1.0, (1.0 - self.min_agree) * own.numel()) + 0.5)
            if not ok:
                note = ''
                if x.shape[2:] == (1, 1) and self.small:        # diagnosis: the nearest handed-over tensor of this shape, anywhere
                    dis = [(int(((e > 0) != own[:, :, 0, 0]).sum()), i) for i, e in enumerate(self.small) if tuple(e.shape) == tuple(x.shape[:2])]
                    if dis:
                        k, i = min(dis)
                        note = f'nearest handed-over tensor: {k} decisions differ at list position {i} (cursor {self.small_pos}); max |x| {float(x.abs().max()):.3e}'
                self.unmatched.append((what, tuple(x.shape), round(best_agree, 4)) + ((note,) if note else ()))
                return None
        dec = best > 0
        diff = dec != own
        self.matched += 1
        self.decisions += own.numel()
        k = int(diff.sum())
        if k:
            self.flips += k
            self.worst_margin = max(self.worst_margin, (x[diff].abs().max() / x.abs().max().clamp_min(1e-30)).item())
        return dec

    def argmax(self, cols: torch.Tensor, x: torch.Tensor, k: int, stride: int, padding: int):
        """window arg-max indices [N,C,L] as the other implementation saw them, or None.  cols: the oracle's unfolded windows
        [N,C,k*k,L].  A window is LIVE when its maximum is positive (it survives the ReLU around the pool; the others carry no
        gradient): liveness, like the arg-max, is read from the candidate — a replayed ReLU decision can leave the oracle's own
        window without a positive entry although the other implementation routes a gradient through it."""
        self.sites += 1
        own = cols.argmax(dim=2)
        own_live = cols.max(dim=2).values > 0
        best, best_agree, best_live = None, 0.0, None
        for e in self._shaped(x):
            ep = F.pad(e, (padding,) * 4, value=float('-inf')) if padding else e
            ecols = F.unfold(ep, k, stride=stride).view(*cols.shape)
            idx, live = ecols.argmax(dim=2), ecols.max(dim=2).values > 0
            both = live & own_live
            agree = (idx == own)[both].float().mean().item() if both.any() else 1.0
            if agree > best_agree:
                best, best_agree, best_live = idx, agree, live
        if best is None or best_agree < self.min_agree:
            self.unmatched.append(('max_pool2d', tuple(x.shape), round(best_agree, 4)))
            return None
        self.matched += 1
        self.decisions += int(best_live.sum())
        diff = (best != own) & best_live
        n_diff = int(diff.sum())
        if n_diff:
            self.flips += n_diff
            top = cols.max(dim=2).values
            taken = cols.gather(2, best.unsqueeze(2)).squeeze(2)
            self.worst_margin = max(self.worst_margin, ((top - taken)[diff].max() / x.abs().max().clamp_min(1e-30)).item())
        return torch.where(best_live, best, own)

    def summary(self) -> str:
        return (f'{self.matched}/{self.sites} kink sites matched, {self.flips} of {self.decisions} decisions differ from the '
                f"oracle's own (all within {self.worst_margin:.1e} of a tie, relative to the tensor's max)")


@contextmanager
def replaying(candidates, min_agree: float = 0.999, small=()):
    old = dict(_STATE)
    rp = Replay(candidates, min_agree, small)
    _STATE.update(flip=False, replay=rp)
    try:
        yield rp
    finally:
        _STATE.clear()
        _STATE.update(old)


@contextmanager
def flipped(delta: float):
    """yields a dict that holds, after the block, how many decisions were flipped (`count`) of how many (`total`)"""
    old = dict(_STATE)
    stats = {}
    _STATE.update(delta=float(delta), flip=True, count=0, total=0)
    try:
        yield stats
    finally:
        stats.update(count=_STATE['count'], total=_STATE['total'])
        _STATE.clear()
        _STATE.update(old)


def _note(near: torch.Tensor):
    _STATE['count'] += int(near.sum())
    _STATE['total'] += near.numel()


def _route(y_value: torch.Tensor, x: torch.Tensor, slope: torch.Tensor) -> torch.Tensor:
    """value of y_value, gradient d/dx = slope"""
    return y_value.detach() + (x - x.detach()) * slope


def _replayed_slope(x, pos_slope, neg_slope, what):
    """values and gradient of a two-slope function under the replayed decisions (None: no match, caller falls back)"""
    if not x.requires_grad:
        # not on the differentiated path (e.g. the mapping network over the latent NOISE of a StyleGAN defender): no gradient is
        # routed through this site, its decisions cannot move d loss / d input, and the other implementation need not keep it
        return None
    xd = x.detach()
    dec = _STATE['replay'].sign(xd, what)
    if dec is None:
        return None
    slope = torch.where(dec, pos_slope, neg_slope)
    return (xd * slope).detach() + (x - xd) * slope


def leaky_relu(x: torch.Tensor, negative_slope: float = 0.01) -> torch.Tensor:
    if _STATE['replay'] is not None:
        one = torch.ones((), dtype=x.dtype)
        y = _replayed_slope(x, one, one * negative_slope, f'leaky_relu({negative_slope})')
        return y if y is not None else F.leaky_relu(x, negative_slope)
    if not _STATE['flip']:
        return F.leaky_relu(x, negative_slope)
    xd = x.detach()
    near = xd.abs() < _STATE['delta']
    _note(near)
    pos = (xd > 0) ^ near
    return _route(F.leaky_relu(xd, negative_slope), x, torch.where(pos, 1.0, negative_slope).to(x.dtype))


def relu(x: torch.Tensor) -> torch.Tensor:
    if _STATE['replay'] is not None:
        return leaky_relu(x, 0.0)
    if not _STATE['flip']:
        return F.relu(x)
    return leaky_relu(x, 0.0)


def prelu(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    if _STATE['replay'] is not None:
        w = weight.view(1, -1, *([1] * (x.ndim - 2))).expand_as(x)
        y = _replayed_slope(x, torch.ones_like(x.detach()), w.detach(), 'prelu')
        return y if y is not None else F.prelu(x, weight)
    if not _STATE['flip']:
        return F.prelu(x, weight)
    xd = x.detach()
    near = xd.abs() < _STATE['delta']
    _note(near)
    pos = (xd > 0) ^ near
    w = weight.view(1, -1, *([1] * (x.ndim - 2)))
    return _route(F.prelu(xd, weight), x, torch.where(pos, torch.ones_like(xd), w.expand_as(xd)))


def clamp(x: torch.Tensor, lo: float, hi: float) -> torch.Tensor:
    if not _STATE['flip']:            # (decision replay keeps the oracle's own clamp decisions)
        return torch.clamp(x, lo, hi)
    xd = x.detach()
    near = ((xd - lo).abs() < _STATE['delta']) | ((xd - hi).abs() < _STATE['delta'])
    _note(near)
    inside = ((xd >= lo) & (xd <= hi)) ^ near
    return _route(torch.clamp(xd, lo, hi), x, inside.to(x.dtype))


def max_pool2d(x: torch.Tensor, kernel_size: int, stride: int, padding: int = 0) -> torch.Tensor:
    if not _STATE['flip'] and _STATE['replay'] is None:
        return F.max_pool2d(x, kernel_size, stride, padding)
    n, c, h, w = x.shape
    k = kernel_size
    xp = F.pad(x, (padding,) * 4, value=float('-inf')) if padding else x
    cols = F.unfold(xp, k, stride=stride).view(n, c, k * k, -1)
    if _STATE['replay'] is not None and not x.requires_grad:
        return F.max_pool2d(x, kernel_size, stride, padding)          # off the differentiated path: see _replayed_slope
    if _STATE['replay'] is not None:
        idx = _STATE['replay'].argmax(cols.detach(), x.detach(), k, stride, padding)
        if idx is None:
            return F.max_pool2d(x, kernel_size, stride, padding)
        y = cols.gather(2, idx.unsqueeze(2)).squeeze(2)
        return y.view(n, c, (h + 2 * padding - k) // stride + 1, -1)
    top = cols.detach().topk(2, dim=2)
    near = (top.values[:, :, 0] - top.values[:, :, 1]) < _STATE['delta']
    _note(near)
    idx = torch.where(near, top.indices[:, :, 1], top.indices[:, :, 0])
    y = cols.gather(2, idx.unsqueeze(2)).squeeze(2)
    ho = (h + 2 * padding - k) // stride + 1
    return y.view(n, c, ho, -1)


def tie_mask(grad_normal: torch.Tensor, grad_flipped: torch.Tensor, tol: float) -> torch.Tensor:
    """True where a near-tie decision can move the gradient element by more than a quarter of the tolerance
    (tol is relative to the gradient's max magnitude, as the parity assertions are)"""
    scale = max(grad_normal.abs().max().item(), 1e-30)
    return (grad_normal - grad_flipped).abs() > 0.25 * tol * scale


def assert_grad_close(got: torch.Tensor, grad_normal: torch.Tensor, grad_flipped: torch.Tensor, tol: float,
                      what: str = '', max_masked: float = 0.2) -> float:
    """|got - grad_normal| <= tol * max|grad| on every element no near-tie decision reaches; returns the masked
    fraction (asserted below `max_masked` so that the check stays meaningful)"""
    scale = max(grad_normal.abs().max().item(), 1e-30)
    mask = tie_mask(grad_normal, grad_flipped, tol)
    frac = mask.float().mean().item()
    err = (got - grad_normal).abs()
    free = err[~mask].max().item() if (~mask).any() else 0.0
    print(f'   {what}: max err off near-ties {free / scale:.2e} (rel. to max |g| {scale:.2e}); '
          f'{frac * 100:.2f} % of the elements depend on a near-tie decision '
          f'(max err there {(err[mask].max().item() / scale if mask.any() else 0.0):.2e})')
    assert frac <= max_masked, f'{what}: {frac:.3f} of the gradient elements are tie-dependent — check delta'
    assert free <= tol * scale, f'{what}: gradient differs by {free / scale:.2e} (rel.) away from near-ties'
    return frac
