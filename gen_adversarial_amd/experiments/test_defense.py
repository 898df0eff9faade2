"""
Multi-GPU evaluation driver: the counterpart of the reference's src/experiments/test_defense.py (one process per GPU,
images sharded across ranks, every image attacked independently, results gathered and merged into results.json).

Kept from the reference: CLI flags (test_defense.py:55-71), `DistributedSampler(shuffle=False)` sharding — rank r takes
indices r::W of the head-padded index list (:116) — rank-major concatenation of the gathered results (:250-253),
results.json keys and the 100.0 = "attack failed" convention (:141-146, 255-291), seeds (:93-97).
Changed for MI355X: the four all_gathers (:245-248) are ONE RCCL all-gather of a (ceil(N/W), 1+#attacks) fp32 tensor;
the per-image barrier (:126-127) is dropped (images are independent; a single barrier precedes the gather);
ranks are launched by torchrun / mp.spawn with MASTER_ADDR=127.0.0.1.
Added (`--schedule dynamic`; SURVEY.md §5 / §8(e): attack cost varies > 100x with early exits, so static `r::W` shards finish at very
different times): a SHARED WORK COUNTER — ranks draw chunks of `--batch_images` images from one atomic counter on the process
group's store (`WorkQueue`), so a rank that drew cheap images simply draws more; still no collective inside the loop, still ONE
all-gather at the end (of the dataset-ordered table, each row filled by the rank that evaluated it).  `--schedule static` stays the
default and the parity mode (the reference's partition and its rank-major row order).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import random
from datetime import timedelta
from typing import Callable, Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

FAILED = 100.0


def shard_indices(n: int, rank: int, world: int) -> List[int]:
    """torch.utils.data.DistributedSampler(shuffle=False, drop_last=False): pad with the head, take rank::world."""
    per = math.ceil(n / world)
    idx = list(range(n))
    total = per * world
    while len(idx) < total:
        idx += idx[:total - len(idx)]
    return idx[rank:total:world]


def seed_everything(seed: int = 42):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def evaluate_shard(defense_model, attacks: Dict[str, Callable], images: torch.Tensor, labels: torch.Tensor,
                   batch_images: int = 1) -> torch.Tensor:
    """per image: [clean correct (0/1), distortion per attack] — the loop of test_defense.py:123-200, one image at a time
    exactly as the reference drives its attacks.  batch_images > 1: the clean pass and every attack that declares
    `batched = True` (PGD-Linf: images are attacked independently, per-image early stop) take `batch_images` images per
    call, i.e. batch_images x EoT defender rows per plan run — the MI355X is launch-bound on one image x EoT 32; the
    reference's single-image attacks keep their one-image protocol."""
    out = torch.zeros(images.shape[0], 1 + len(attacks))
    if batch_images > 1:
        per_image = {j: a for j, a in enumerate(attacks.values()) if not getattr(a, 'batched', False)}
        for lo in range(0, images.shape[0], batch_images):
            x = images[lo:lo + batch_images].clamp(0.0, 1.0)
            y = labels[lo:lo + batch_images]
            with torch.no_grad():
                out[lo:lo + x.shape[0], 0] = (defense_model(x).argmax(dim=1) == y).float().cpu()
            for j, attack in enumerate(attacks.values()):
                if j in per_image:
                    continue
                success, bound, _ = attack(x, y, defense_model)
                success = torch.as_tensor(success).view(-1).cpu()
                bound = torch.as_tensor(bound, dtype=torch.float32).view(-1).cpu()
                out[lo:lo + x.shape[0], 1 + j] = torch.where(success, bound, torch.full_like(bound, FAILED))
        for i in range(images.shape[0]):
            for j, attack in per_image.items():
                success, bound, _ = attack(images[i:i + 1].clamp(0.0, 1.0), labels[i:i + 1], defense_model)
                out[i, 1 + j] = float(bound) if success else FAILED
        return out
    for i in range(images.shape[0]):
        x = images[i:i + 1].clamp(0.0, 1.0)
        y = labels[i:i + 1]
        with torch.no_grad():
            out[i, 0] = float((defense_model(x).argmax(dim=1) == y).item())
        for j, (_, attack) in enumerate(attacks.items()):
            success, bound, _ = attack(x, y, defense_model)
            out[i, 1 + j] = float(bound) if success else FAILED
    return out


class WorkQueue:
    """chunk indices 0, 1, 2, ... handed out once each across all ranks: one atomic add on the process group's key-value store per
    chunk (TCPStore.add is atomic; ~100 us against seconds of attack per chunk).  World size 1: a plain counter."""

    def __init__(self, world: int, tag: str = 'eval'):
        self.key, self.local, self.store = f'ga_work_{tag}', 0, None
        if world > 1:
            from torch.distributed import distributed_c10d as c10d
            self.store = c10d._get_default_store()

    def next(self) -> int:
        if self.store is None:
            self.local += 1
            return self.local - 1
        return int(self.store.add(self.key, 1)) - 1


def evaluate_dynamic(defense_model, attacks: Dict[str, Callable], images: torch.Tensor, labels: torch.Tensor, world: int,
                     batch_images: int = 1, device=None, queue: 'WorkQueue' = None):
    """the whole dataset through the shared work counter: returns (table [N, 1 + 1 + #attacks] with column 0 = "this rank evaluated the
    row", busy seconds of this rank, images this rank evaluated)"""
    import time
    n = images.shape[0]
    chunk = max(1, int(batch_images))
    queue = queue or WorkQueue(world)
    table = torch.zeros(n, 2 + len(attacks))
    busy, done = 0.0, 0
    while True:
        c = queue.next()
        lo = c * chunk
        if lo >= n:
            break
        hi = min(n, lo + chunk)
        t = time.time()
        x, y = images[lo:hi], labels[lo:hi]
        if device is not None:
            x, y = x.to(device), y.to(device)
        table[lo:hi, 1:] = evaluate_shard(defense_model, attacks, x, y, batch_images=chunk)
        table[lo:hi, 0] = 1.0
        busy += time.time() - t
        done += hi - lo
    return table, busy, done


def gather_dynamic(table: torch.Tensor, world: int, device) -> torch.Tensor:
    """ONE all-gather of the dataset-ordered tables; every row was filled by exactly one rank (column 0 says which)"""
    if world == 1:
        assert bool((table[:, 0] == 1).all())
        return table[:, 1:]
    table = table.to(device)
    parts = [torch.zeros_like(table) for _ in range(world)]
    dist.barrier()
    dist.all_gather(parts, table)
    total = torch.stack(parts).sum(dim=0).cpu()
    if not bool((total[:, 0] == 1).all()):
        raise RuntimeError('work queue: some image was evaluated by no rank or by several')
    return total[:, 1:]


def gather_results(local: torch.Tensor, world: int, device) -> torch.Tensor:
    """ONE all-gather of the whole per-image table (payload: a few KB -> latency bound on xGMI)."""
    if world == 1:
        return local
    local = local.to(device)
    parts = [torch.zeros_like(local) for _ in range(world)]
    dist.barrier()
    dist.all_gather(parts, local)
    return torch.cat(parts, dim=0).cpu()                       # rank-major, like test_defense.py:250-253


# results.json column of each attack (test_defense.py:267-287 writes 'Clean', 'DeepFool', 'C&W', 'AutoAttack'; 'PGD' is this
# build's plug-in and has no reference counterpart)
RESULT_KEYS = {'deepfool': 'DeepFool', 'c&w': 'C&W', 'autoattack': 'AutoAttack', 'pgd': 'PGD', 'pgd-bpda': 'PGD-BPDA'}


def merge_results(path: str, clean: float, columns: Dict[str, List[float]]):
    """read-modify-write of results.json (test_defense.py:255-291)."""
    res = {}
    if os.path.exists(path):
        with open(path) as f:
            res = json.load(f)
    res['Clean'] = clean
    res.update(columns)
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    with open(path, 'w') as f:
        json.dump(res, f)
    return res


def robust_accuracy(distortions: Sequence[float], eps: float) -> float:
    """robust-acc@eps = mean(distortion > eps) with failure = 100 and clean-misclassified = 0."""
    d = np.asarray(distortions, dtype=np.float64)
    return float((d > eps).mean()) if d.size else float('nan')


def run_worker(rank: int, world: int, args, make_model: Callable, dataset: Tuple[torch.Tensor, torch.Tensor],
               backend: str = 'nccl', results_path: str = None):
    """one rank: shard, evaluate, gather, merge.  `make_model(args) -> (args, defense_model)` is load_defense.load."""
    seed_everything(42)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '12355')
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=timedelta(hours=12))
    args, defense_model = make_model(args)
    images, labels = dataset
    dev = args.device
    bi = int(getattr(args, 'batch_images', 1) or 1)
    if getattr(args, 'schedule', 'static') == 'dynamic':
        local, busy, done = evaluate_dynamic(defense_model, args.attacks, images, labels, world, batch_images=bi, device=dev)
        args.rank_busy_seconds, args.rank_images = busy, done          # (tests and logs read these)
        table = gather_dynamic(local, world, dev)
    else:
        mine = shard_indices(images.shape[0], rank, world)
        local = evaluate_shard(defense_model, args.attacks, images[mine].to(dev), labels[mine].to(dev), batch_images=bi)
        table = gather_results(local, world, dev)
    res = None
    if rank == 0:
        cols = {RESULT_KEYS.get(name, name): table[:, 1 + j].tolist() for j, name in enumerate(args.attacks.keys())}
        res = merge_results(results_path or os.path.join(args.results_folder, 'results.json'),
                            float(table[:, 0].mean().item()), cols)
    if world > 1:
        dist.barrier()
    return res


def synthetic_dataset(n: int, size: int, n_classes: int, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, 3, size, size, generator=g), torch.randint(0, n_classes, (n,), generator=g)


IMAGE_PATTERNS = ('*.png', '*.jpg', '*.bmp', '*.JPEG')


def folder_dataset(folder: str, size: int):
    """ImageLabelDataset (data/datasets.py:35-58): every *.png / *.jpg / *.bmp / *.JPEG below `folder` (recursive), sorted by
    path; label = index of the image's parent directory name among the sorted names of the directories that HOLD images.
    I/O is outside the accelerated path and kept minimal (PIL + antialiased bilinear resize = ToTensor + Resize(antialias))."""
    import pathlib
    from PIL import Image
    root = pathlib.Path(folder)
    samples = sorted(p for pat in IMAGE_PATTERNS for p in root.rglob(pat))
    samples = [p.absolute().as_posix() for p in samples]
    if not samples:
        raise FileNotFoundError(f'no {"/".join(IMAGE_PATTERNS)} images below {folder}')
    labels_as_str = [p.split('/')[-2] for p in samples]
    class_names = sorted(set(labels_as_str))
    xs = []
    for f in samples:
        im = Image.open(f).convert('RGB')
        t = torch.from_numpy(np.asarray(im, dtype=np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
        t = torch.nn.functional.interpolate(t, size=(size, size), mode='bilinear', antialias=True, align_corners=False)
        xs.append(t[0])
    return torch.stack(xs), torch.tensor([class_names.index(s_) for s_ in labels_as_str])


def parse_args(argv=None):
    p = argparse.ArgumentParser('Common Pipeline to test a given defense mechanism.')
    p.add_argument('--images_path', type=str, default=None, help='All images in this folder will be attacked')
    p.add_argument('--synthetic', type=int, default=0, help='use N synthetic images instead of --images_path')
    p.add_argument('--eot_steps', type=int, default=32)
    p.add_argument('--defense_type', type=str, choices=['base', 'A-VAE', 'ND-VAE', 'trades', 'ours', 'ablation'])
    p.add_argument('--experiment', type=str, choices=['gender', 'ids', 'cars'])
    p.add_argument('--config', type=str, required=True)
    p.add_argument('--attack', type=str, choices=['deepfool', 'c&w', 'autoattack', 'pgd', 'pgd-bpda'], default=None,
                   help='If passed, try a specific attack only. Otherwise, try all (the reference\'s three).')
    p.add_argument('--batch_images', type=int, default=1,
                   help='images per defender call for the clean pass and for batched attacks (PGD); 1 = the reference protocol')
    p.add_argument('--schedule', type=str, choices=['static', 'dynamic'], default='static',
                   help='static: the reference\'s r::W partition (parity mode); dynamic: ranks draw chunks of --batch_images images '
                        'from a shared work counter (balances attacks whose cost varies with early exits)')
    args = p.parse_args(argv)
    args.results_folder = f'./results/{args.config.split("/")[-1][:-5]}/'
    os.makedirs(args.results_folder, exist_ok=True)
    return args


def main():
    from .load_defense import load
    args = parse_args()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    args.device = f'cuda:{local_rank}'

    def make_model(a):
        a, m = load(a)
        if a.attack == 'pgd':
            a.attacks = {'pgd': a.pgd}
        elif a.attack == 'pgd-bpda':
            a.attacks = {'pgd-bpda': getattr(a, 'pgd_bpda', None) or type(a.pgd)(eps=a.pgd.eps, step_size=a.pgd.step_size, steps=a.pgd.steps, bpda=True)}
        elif a.attack is not None:
            a.attacks = {k: v for k, v in a.attacks.items() if k == a.attack}
        return a, m
    size = {'ids': 64, 'gender': 256, 'cars': 128}[args.experiment]
    data = synthetic_dataset(args.synthetic, size, 100) if args.synthetic else folder_dataset(args.images_path, size)
    res = run_worker(rank, world, args, make_model, data, backend='nccl')
    if rank == 0:
        print(json.dumps({k: (v if not isinstance(v, list) else f'{len(v)} values') for k, v in res.items()}))


if __name__ == '__main__':
    main()
