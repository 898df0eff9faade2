"""Bare bf16 MFMA loops on the two gfx950 shapes, interleaved rounds in one process (guide rule 24): does the chip hold a higher
clock on v_mfma_f32_16x16x32_bf16 than on 32x32x16 (MI355X_MICROARCH.md, DVFS give-back item 7)?  Prints one JSON line."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gen_adversarial_amd import _lib as L

dev = torch.device('cuda:0')
blocks, iters = 256 * 4, 20000
out = torch.empty(blocks * 256, device=dev)
st = torch.cuda.current_stream().cuda_stream
flops = blocks * 4 * iters * 8 * 2 * 32 * 32 * 16


def run(shape, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        L.check(L.lib.ga_microbench_mfma_bf16_shape(out.data_ptr(), blocks, iters, shape, st), 'mfma')
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


for s in (32, 16):          # warm the chip: ~2 s of back-to-back launches
    run(s, 20)
res = {16: [], 32: []}
for r in range(6):
    for s in (32, 16):
        res[s].append(flops / (run(s, 5) * 1e-3) / 1e12)
print(json.dumps({'what': 'bare bf16 MFMA loop, TFLOP/s per round, interleaved', 'shape_32x32x16': [round(v, 1) for v in res[32]],
                  'shape_16x16x32': [round(v, 1) for v in res[16]],
                  'median_ratio_16_over_32': round(sorted(res[16])[3] / sorted(res[32])[3], 4)}))
