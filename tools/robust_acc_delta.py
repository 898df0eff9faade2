"""GPU tool: robust-accuracy delta of the HIP path against the CPU oracle at a sample size of your choice (tests/robust_acc.py is
the harness; bench.py reports the 512-image figure in every run).  TEST INFRASTRUCTURE: the oracle is the checker here.

    python tools/robust_acc_delta.py [images=2048] [eot=4] [pgd_steps=6]        prints one JSON line (PGD-Linf, tests/robust_acc.py)
    python tools/robust_acc_delta.py apgd [images=4096] [eot=2] [n_iter=5] [l2_bound=2.0]
        the reference's APGD-CE at a fixed L2 bound with a paired 95 % interval (tests/robust_acc_attack.py); a progress line per
        chunk of 64 images keeps a long run visibly alive
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from robust_acc import robust_accuracy_delta   # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == 'apgd':
    from robust_acc_attack import robust_accuracy_under_attack
    a = sys.argv[2:]
    n = int(a[0]) if len(a) > 0 else 4096
    print(json.dumps(robust_accuracy_under_attack('cuda:0', n_images=n, eot=int(a[1]) if len(a) > 1 else 2,
                                                  n_iter=int(a[2]) if len(a) > 2 else 5, bound=float(a[3]) if len(a) > 3 else 2.0,
                                                  chunk_images=min(64, n), progress=True)), flush=True)
    sys.exit(0)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
eot = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
print(json.dumps(robust_accuracy_delta('cuda:0', n_images=n, eot=eot, steps=steps, chunk_images=min(128, n))), flush=True)
