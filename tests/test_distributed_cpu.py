"""CPU, world_size 2 over gloo: the sharded driver gives the same table as one rank, gathered with ONE collective."""
import json
import os
from argparse import Namespace

import pytest
import torch
import torch.multiprocessing as mp

from gen_adversarial_amd.attacks.pgd import PGDLinf
from gen_adversarial_amd.experiments import test_defense as drv


def _make_model(args):
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(3 * 8 * 8, 5)).eval()
    args.attacks = {'pgd': PGDLinf(eps=8 / 255, step_size=2 / 255, steps=5)}
    return args, net


def _worker(rank, world, port, path, n):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    args = Namespace(device='cpu', results_folder=os.path.dirname(path))
    data = drv.synthetic_dataset(n, 8, 5, seed=3)
    drv.run_worker(rank, world, args, _make_model, data, backend='gloo', results_path=path)
    if world > 1:
        torch.distributed.destroy_process_group()


def test_shard_indices_follow_distributed_sampler():
    from torch.utils.data import DistributedSampler
    for n, w in ((10, 4), (7, 2), (8, 8), (3, 4), (1, 2)):
        for r in range(w):
            ref = list(DistributedSampler(list(range(n)), num_replicas=w, rank=r, shuffle=False))
            assert drv.shard_indices(n, r, w) == ref, (n, w, r)


@pytest.mark.parametrize('n', [6, 5])
def test_two_ranks_equal_one_rank(tmp_path, n):
    p1, p2 = str(tmp_path / 'one' / 'results.json'), str(tmp_path / 'two' / 'results.json')
    _worker(0, 1, 0, p1, n)
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, p2, n), nprocs=2, join=True)
    r1, r2 = json.load(open(p1)), json.load(open(p2))
    # rank-major order with head padding (reference semantics): compare as per-image tables
    order = drv.shard_indices(n, 0, 2) + drv.shard_indices(n, 1, 2)
    assert len(r2['pgd']) == len(order)
    for pos, img in enumerate(order):
        assert abs(r2['pgd'][pos] - r1['pgd'][img]) < 1e-6
    if n % 2 == 0:
        assert abs(r1['Clean'] - r2['Clean']) < 1e-6
    assert 0.0 <= drv.robust_accuracy(r2['pgd'], 4 / 255) <= 1.0


def test_merge_results_is_read_modify_write(tmp_path):
    p = str(tmp_path / 'results.json')
    drv.merge_results(p, 0.5, {'DeepFool': [1.0, 100.0]})
    r = drv.merge_results(p, 0.75, {'C&W': [0.5]})
    assert r == {'Clean': 0.75, 'DeepFool': [1.0, 100.0], 'C&W': [0.5]}
    assert drv.robust_accuracy([0.0, 0.5, 100.0], 0.1) == pytest.approx(2 / 3)
