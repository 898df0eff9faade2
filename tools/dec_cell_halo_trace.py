"""GPU tool (debug build only): where the clocks of ga_dec_cell_halo's backward kernel go.
    make -C gen_adversarial_amd/csrc hctrace
    GA_OPS_LIB=gen_adversarial_amd/libga_ops_hctrace.so python tools/dec_cell_halo_trace.py [rows]"""
import ctypes as C
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

sys.argv = [sys.argv[0]] + (sys.argv[1:] or ['64'])
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dec_cell_halo_time.py'), run_name='__main__')
lib = C.CDLL(os.environ['GA_OPS_LIB'])
buf = (C.c_ulonglong * 64)()
assert lib.ga_hc_trace_read(buf, 64) == 0
t = np.array(list(buf), dtype=np.float64).reshape(4, 16)
names = ['setup (operand loads, masks)', 'weights -> LDS, barrier', '(a) GEMM1, SiLU -> PA / P4', 'barrier', '(b) depthwise x2, SiLU\'(t2)', '(c) GEMM3',
         'barrier', 'g -> PA', 'barrier', '(d) * SiLU\'(t2)', 'barrier', '(e) depthwise^T, * P4, split', 'barrier', '(f) GEMM4', 'barrier', 'whole kernel']
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]
print('backward kernel of the LAST shape, workgroup 0, clocks summed over the chunks, per wave:')
for i in order:
    print(f'  {names[i]:38s} ' + ' '.join(f'{v:9.0f}' for v in t[:, i]))
