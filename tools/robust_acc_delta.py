"""GPU tool: robust-accuracy delta of the HIP path against the CPU oracle at a sample size of your choice (tests/robust_acc.py is
the harness; bench.py reports the 512-image figure in every run).  TEST INFRASTRUCTURE: the oracle is the checker here.

    python tools/robust_acc_delta.py [images=2048] [eot=4] [pgd_steps=6]        prints one JSON line (PGD-Linf, tests/robust_acc.py)
    python tools/robust_acc_delta.py apgd [images=4096] [eot=2] [n_iter=5] [l2_bound=2.0] [seed=0]      (pool FILES...: pooled interval)
        the reference's APGD-CE at a fixed L2 bound with a paired 95 % interval (tests/robust_acc_attack.py); a progress line per
        chunk of 64 images keeps a long run visibly alive
    python tools/robust_acc_delta.py fullsize [images=8] [eot=2] [n_iter=3] [l2_bound=2.0]
        same-input verdicts on the FULL-SIZE model: APGD-CE on the HIP engine, clean and adversarial images judged by the engine and
        by the CPU oracle under the same pinned noise (the oracle needs ~1.5 s per row and pass)
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
                                                  seed=int(a[4]) if len(a) > 4 else 0, chunk_images=min(64, n), progress=True)), flush=True)
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == 'pool':
    # pool several apgd runs on the SAME random model (model_seed 0) with different data seeds (images, start noise, latent draws):
    # the paired interval on the pooled verdict pairs
    import math
    runs = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sys.argv[2:]]
    n = sum(r['images'] for r in runs)
    n10, n01 = sum(r['robust_on_hip_only'] for r in runs), sum(r['robust_on_oracle_only'] for r in runs)
    delta = (n10 - n01) / n
    se = math.sqrt(max(n10 + n01 - (n10 - n01) ** 2 / n, 0.0)) / n
    print(json.dumps({'attack': runs[0]['attack'], 'l2_bound': runs[0]['l2_bound'], 'n_iter': runs[0]['n_iter'], 'eot': runs[0]['eot'],
                      'runs': len(runs), 'images': n,
                      'robust_acc_hip': sum(r['robust_acc_hip'] * r['images'] for r in runs) / n,
                      'robust_acc_oracle': sum(r['robust_acc_oracle'] * r['images'] for r in runs) / n,
                      'delta': delta, 'ci95_halfwidth': 1.96 * se, 'ci95': [delta - 1.96 * se, delta + 1.96 * se],
                      'within_0.1_percent': abs(delta) + 1.96 * se <= 1e-3, 'discordant_pairs': n10 + n01,
                      'robust_on_hip_only': n10, 'robust_on_oracle_only': n01,
                      'same_input_verdicts_differing': sum(r['same_input_verdicts_differing'] for r in runs),
                      'same_input_verdict_pairs': sum(r['same_input_verdict_pairs'] for r in runs),
                      'oracle_seconds': sum(r['oracle_seconds'] for r in runs), 'hip_seconds': sum(r['hip_seconds'] for r in runs),
                      'what': 'pooled over runs on the same random model with different data seeds (images, start noise, latent draws): ' + runs[0]['what']}))
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == 'fullsize':
    from robust_acc_attack import fullsize_same_input_verdicts
    a = sys.argv[2:]
    print(json.dumps(fullsize_same_input_verdicts('cuda:0', n_images=int(a[0]) if len(a) > 0 else 8, eot=int(a[1]) if len(a) > 1 else 2,
                                                  n_iter=int(a[2]) if len(a) > 2 else 3, bound=float(a[3]) if len(a) > 3 else 2.0)), flush=True)
    sys.exit(0)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
eot = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
print(json.dumps(robust_accuracy_delta('cuda:0', n_images=n, eot=eot, steps=steps, chunk_images=min(128, n))), flush=True)
