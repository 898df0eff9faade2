"""GPU tool (debug build only): where the clocks of ga_dec_cell's chunk loop go, and its launch time.

    make -C gen_adversarial_amd/csrc dctrace
    GA_OPS_LIB=gen_adversarial_amd/libga_ops_dctrace.so python tools/dec_cell_trace.py N H C [Hd]

Per phase: shader-clock sums over the chunks (lane 0 of each wave of workgroup 0), printed per chunk.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gen_adversarial_amd import _lib as L

N, H, Cc = (int(v) for v in sys.argv[1:4])
Hd = int(sys.argv[4]) if len(sys.argv) > 4 else 6 * Cc
dev = 'cuda'
x = torch.randn(N, H, H, Cc, device=dev)


def split(w):
    hi = w.to(torch.bfloat16)
    return hi.contiguous(), (w - hi.float()).to(torch.bfloat16).contiguous()


w1, w2, w2t = torch.randn(Hd, Cc, device=dev) / Cc ** 0.5, torch.randn(Cc, Hd, device=dev) / Hd ** 0.5, None
w2t = w2.t().contiguous()
(w1h, w1l), (w2h, w2l), (w2th, w2tl) = split(w1), split(w2), split(w2t)
wd, wdb = torch.randn(25, Hd, device=dev) * 0.2, torch.randn(25, Hd, device=dev) * 0.2
b1, bd, b2 = torch.randn(Hd, device=dev) * 0.3, torch.randn(Hd, device=dev) * 0.3, torch.randn(Cc, device=dev) * 0.3
dout, ps, pb = torch.randn(N, H, H, Cc, device=dev), torch.rand(N, Cc, device=dev), torch.randn(N, Cc, device=dev) * 0.01
y, dt1 = torch.empty(N, H, H, Cc, device=dev), torch.empty(N, H, H, Hd, device=dev)
lib = C.CDLL(os.environ['GA_OPS_LIB']) if os.environ.get('GA_OPS_LIB') else None

FWD = ['A depthwise (+ issue next weights)', 'B GEMM1(ch+1) || SiLU + split -> P2', 'barrier 1, W1 / taps -> LDS',
       'C GEMM2(ch) || SiLU(t1(ch+1)) -> P1', 'barrier 2, W2 -> LDS']
BWD = ['weights / taps -> LDS, barrier', 'issue next weights, GEMM1, SiLU / SiLU\' -> P1, P4', 'barrier', 'depthwise', 'SiLU\'(t2) || GEMM3', 'barrier',
       'g -> P1, barrier', '* SiLU\'(t2), barrier', 'depthwise^T, * P4 -> HBM', 'barrier']

for backward in (0, 1):
    d = L.DecCellDesc()
    d.x, d.w1_hi, d.w1_lo, d.b1 = x.data_ptr(), w1h.data_ptr(), w1l.data_ptr(), b1.data_ptr()
    d.wd, d.wd_bwd, d.bd, d.b2 = wd.data_ptr(), wdb.data_ptr(), bd.data_ptr(), b2.data_ptr()
    if backward:
        d.w2_hi, d.w2_lo, d.y = w2th.data_ptr(), w2tl.data_ptr(), dt1.data_ptr()
        d.dout, d.pro_scale, d.pro_shift = dout.data_ptr(), ps.data_ptr(), pb.data_ptr()
    else:
        d.w2_hi, d.w2_lo, d.y = w2h.data_ptr(), w2l.data_ptr(), y.data_ptr()
    d.N, d.H, d.W, d.C, d.Hd, d.backward = N, H, H, Cc, Hd, backward
    d.variant = int(os.environ.get('GA_DEC_CELL_VARIANT', '0'))
    for _ in range(3):
        L.run(d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.run(d)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    M = 256 if Cc == 128 else 128
    wgs = N * H * H // M
    print(f'{"backward" if backward else "forward"}: {us:.1f} us, {wgs} workgroups, {Hd // 32} chunks')
    if lib is not None:
        buf = np.zeros(64, dtype=np.uint64)
        assert lib.ga_debug_dc_trace_read(buf.ctypes.data_as(C.c_void_p), C.c_int(64)) == 0
        names = BWD if backward else FWD
        t = buf.reshape(4, 16)[:, :len(names)].astype(np.float64) / (Hd // 32)
        tot = t.sum(axis=1)
        for i, nm in enumerate(names):
            print(f'   {nm:34s} ' + ' '.join(f'{v:8.0f}' for v in t[:, i]) + f'   ({100 * t[:, i].mean() / tot.mean():4.1f} %)')
        print(f'   {"per chunk":34s} ' + ' '.join(f'{v:8.0f}' for v in tot))
