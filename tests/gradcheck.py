"""Gradient parity at the kinks of the path (oracle/kinks.py).

A 1-ulp difference in a forward value can flip a ReLU / LeakyReLU / PReLU / max-pool decision and move gradient
contributions by O(1).  The parity statement that can hold exactly is therefore two-fold, and both halves are asserted:
  (a) GIVEN the decisions the HIP engine took (read from its own stored activations and replayed in the oracle), the engine's
      gradient equals the oracle's on EVERY element at the path's tolerance (1e-3 of max |g|; observed ~1e-5);
  (b) the engine's decisions differ from the oracle's own only within `max_margin` of a tie (relative to the tensor's max).
`tie-mask` form (oracle_grads / assert_grad_close): the elements on which flipping every near-tie decision changes the
oracle's gradient are masked — sharp for single layers, nearly vacuous for networks with a global receptive field."""
import torch

from oracle import kinks as K

# near-tie width per engine precision for the tie-mask form: ~10x the observed forward difference HIP vs CPU oracle
DELTA = {'fp32': 3e-5, 'bf16x3': 3e-4}


def oracle_grads(loss_fn, x: torch.Tensor, delta: float):
    """loss_fn(x) -> scalar.  Returns (grad, grad with near-tie decisions flipped)."""
    xr = x.detach().clone().requires_grad_(True)
    (g,) = torch.autograd.grad(loss_fn(xr), [xr])
    with K.flipped(delta):
        xf = x.detach().clone().requires_grad_(True)
        (gf,) = torch.autograd.grad(loss_fn(xf), [xf])
    return g, gf


def assert_grad_close(got, g, gf, tol, what='input gradient', max_masked=0.2):
    return K.assert_grad_close(got.detach().cpu().float(), g, gf, tol, what, max_masked)


def engine_candidates(eng, rows=None):
    """every stored activation of the engine ([N,H,W,C] device tensors) as NCHW CPU tensors; rows: a slice of the engine's rows
    (an oracle run on a few rows of a large plan: rows are independent)"""
    out = []
    for a in eng.acts.values():
        if a.t.dim() != 4:
            continue
        t = a.t.detach()
        if rows is not None and t.shape[0] == eng.rows:
            t = t[rows]
        out.append(t.permute(0, 3, 1, 2).float().cpu())
    extra = getattr(eng, 'extra_kink_tensors', None)            # decision tensors the engine does not store as such (A-VAE: conv + noise)
    for t in (extra() if extra is not None else []):
        t = t.detach()
        if rows is not None and t.shape[0] == eng.rows:
            t = t[rows]
        out.append(t.permute(0, 3, 1, 2).float().cpu())
    return out


def assert_grad_given_engine_decisions(eng, loss_fn, x, got, tol=1e-3, what='input gradient', max_margin=1e-4,
                                       min_matched=1, golden=None, rows=None, allow_unmatched=0):
    """(a) + (b) above.  loss_fn(x) -> scalar evaluates the ORACLE; eng is the engine whose last forward produced `got`.
    golden: optional (reference gradient from a golden file, the oracle's own gradient computed HERE).  The two agree at 1e-5
    where the golden was made (tests/test_oracle_golden.py); on another CPU the oracle can decide a near-tie the other way
    (different summation order), which moves that SAMPLE's gradient.  Samples (rows) on which this machine's oracle still
    reproduces the golden are compared against the reference's numbers; the others are reported and covered by the
    oracle-replay comparison alone."""
    small = [t.detach() for t in getattr(eng, 'small_kinks', [])]                    # SE hidden pre-activations, in call order
    # an encoder shared by the EoT replicas keeps one row per IMAGE: the oracle's literal repeat sees it once per replica
    small = [t.repeat_interleave(eng.rows // t.shape[0], dim=0) if (t.shape[0] != eng.rows and eng.rows % t.shape[0] == 0) else t
             for t in small]
    small = [(t[rows] if rows is not None and t.shape[0] == eng.rows else t).float().cpu() for t in small]
    with K.replaying(engine_candidates(eng, rows), small=small) as rp:
        xr = x.detach().clone().requires_grad_(True)
        (g,) = torch.autograd.grad(loss_fn(xr), [xr])
    got = got.detach().cpu().float()
    scale = max(g.abs().max().item(), 1e-30)
    err = (got - g).abs().max().item() / scale
    print(f'   {what}: max err {err:.2e} of max |g| {scale:.2e} given the engine\'s decisions; {rp.summary()}')
    # A site of >= 48 decisions that found no engine tensor keeps the ORACLE's decisions: the comparison below would then no longer be
    # "given the engine's decisions" there.  That is a failure, not a remark: `allow_unmatched` is the
    # number of such sites a caller can justify (tensors the engine provably does not keep), 0 by default.  Sites with fewer
    # decisions (SE hidden units) are handed over in call order (`small`) and only listed when they miss.
    big = [u for u in rp.unmatched if torch.Size(u[1]).numel() >= 48]
    if rp.unmatched:
        print(f'      {len(rp.unmatched)} unmatched sites keep the oracle\'s decisions; of >= 48 decisions: {big[:8]}')
    assert len(big) <= allow_unmatched, (f'{what}: {len(big)} kink sites of >= 48 decisions matched no engine activation '
                                         f'(allowed {allow_unmatched}): {big[:8]}')
    assert rp.matched >= min_matched, f'{what}: only {rp.matched} kink sites matched to engine activations'
    assert rp.worst_margin <= max_margin, f'{what}: a decision flipped {rp.worst_margin:.2e} (relative) away from its tie'
    assert err <= tol, f'{what}: {err:.2e} (relative to max |g|) with the engine\'s own decisions replayed'
    if golden is not None:
        ref, g0 = golden
        rscale = max(ref.abs().max().item(), 1e-30)
        row_ok = ((g0 - ref).flatten(1).abs().amax(dim=1) <= 1e-4 * rscale)
        shifted = g + (ref - g0)                  # the replayed gradient in the reference's numbers
        e_rows = (got - shifted).flatten(1).abs().amax(dim=1) / rscale
        print(f'      vs the reference golden: rows reproduced by this CPU\'s oracle {row_ok.tolist()}, err per row {[f"{v:.1e}" for v in e_rows.tolist()]}')
        assert row_ok.any(), f'{what}: this machine\'s oracle reproduces no row of the golden gradient'
        assert e_rows[row_ok].max().item() <= tol, f'{what}: differs from the reference golden'
    return rp
