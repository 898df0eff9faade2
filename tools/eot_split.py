"""GPU tool: the reference's attack protocol (ONE image x EoT 32 per defender call, src/experiments/test_defense.py:116,133-182) with
the 32 EoT replicas split over S plans of 32 / S rows on S HIP streams: at this size every launch is bound by its own latency and
by the gap between dependent launches of one queue, so S independent queues fill each other's gaps.  The replicas are independent
until the EoT mean, so the split is exact: forward on all streams, join, d CE / d mean-logits, backward on all streams, join, sum
the S input gradients.

    python tools/eot_split.py [S ...]          default 1 2 4
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model

EOT = 32
DEV = 'cuda:0'


class SplitStep:
    def __init__(self, S, model_store=None):
        from gen_adversarial_amd.engine import Engine
        from gen_adversarial_amd.nvae_spec import ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION
        rows = EOT // S
        e0, model = build_model(DEV, rows, rows, seed=0, store=model_store)
        sd, vsd, vspec, alphas = model
        self.engines = [e0] + [Engine(sd, ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, vsd, vspec, rows=rows, rep=rows, alphas=alphas,
                                      temperature=0.6, noise_eps=e0.noise_eps, device=DEV, store=e0.store) for _ in range(S - 1)]
        self.streams = [torch.cuda.Stream(device=DEV) for _ in range(S)]
        self.S = S
        g = torch.Generator(device=DEV).manual_seed(3)
        self.x = torch.rand(1, 3, 64, 64, device=DEV, generator=g)
        self.x_adv = self.x.clone()
        self.label = torch.zeros(1, dtype=torch.long, device=DEV)
        self.part = torch.zeros(S, 100, device=DEV)
        self.gsum = torch.zeros(S, 1, 3, 64, 64, device=DEV)

    def __call__(self):
        main = torch.cuda.current_stream()
        for s in self.streams:
            s.wait_stream(main)
        for k, (e, s) in enumerate(zip(self.engines, self.streams)):
            with torch.cuda.stream(s):
                e.x_in.copy_(self.x_adv)
                for b in e.eps:
                    b.normal_()
                e.forward()
                self.part[k] = e.logits.view(e.rows, -1).sum(dim=0)
        for s in self.streams:
            main.wait_stream(s)
        logits = self.part.sum(dim=0, keepdim=True) / EOT                       # EoT mean over all 32 replicas
        p = torch.softmax(logits, dim=1)
        p[0, self.label] -= 1.0
        for s in self.streams:
            s.wait_stream(main)
        for k, (e, s) in enumerate(zip(self.engines, self.streams)):
            with torch.cuda.stream(s):
                e.dlogits.view(e.rows, -1).copy_((p / EOT).expand(e.rows, -1))
                e.backward()
                self.gsum[k] = e.dx
        for s in self.streams:
            main.wait_stream(s)
        g = self.gsum.sum(dim=0)
        nxt = self.x_adv + (2.0 / 255.0) * g.sign()
        self.x_adv.copy_(torch.min(torch.max(nxt, self.x - 8.0 / 255.0), self.x + 8.0 / 255.0).clamp_(0.0, 1.0))
        return logits


def timed(fn, n, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


if __name__ == '__main__':
    splits = [int(v) for v in sys.argv[1:]] or [1, 2, 4]
    store = None
    for S in splits:
        st = SplitStep(S, store)
        store = st.engines[0].store
        t = timed(st, 20)
        for e, s_ in zip(st.engines, st.streams):              # the same plans replayed as HIP graphs (no host cost per launch)
            with torch.cuda.stream(s_):
                e.enable_graphs()
        tg = timed(st, 20)
        print(f'1 image x EoT {EOT} as {S} plan(s) of {EOT // S} rows on {S} stream(s): {t * 1e3:.2f} ms per attack step = {EOT / t:.0f} rows/s eager, '
              f'{tg * 1e3:.2f} ms = {EOT / tg:.0f} rows/s as HIP graphs ({len(st.engines[0].fwd) + len(st.engines[0].bwd)} launches per plan)', flush=True)
        del st
        import gc
        gc.collect()
        torch.cuda.empty_cache()
