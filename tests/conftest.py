import ast
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_cfg(g):
    cfg = {str(k): ast.literal_eval(str(v)) for k, v in zip(g['cfg_keys'], g['cfg_vals'])}
    res = tuple(int(v) for v in g['res'])
    return cfg, res


@pytest.fixture(scope='session')
def golden_cases():
    return {n: load_golden(f'nvae_{n}.npz') for n in ('A_cos07', 'A_zero_noise2', 'B_adaptive', 'A_nf2')}
