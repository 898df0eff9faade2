"""GPU tool: robust-accuracy delta of the HIP path against the CPU oracle at a sample size of your choice (tests/robust_acc.py is
the harness; bench.py reports the 512-image figure in every run).  TEST INFRASTRUCTURE: the oracle is the checker here.

    python tools/robust_acc_delta.py [images=2048] [eot=4] [pgd_steps=6]        prints one JSON line
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from robust_acc import robust_accuracy_delta   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
eot = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
print(json.dumps(robust_accuracy_delta('cuda:0', n_images=n, eot=eot, steps=steps, chunk_images=min(128, n))), flush=True)
