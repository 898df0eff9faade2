"""
bench.py — purification-under-attack throughput on MI355X (contract: see the task statement / DESIGN.md §Measurement).

One "step" = one white-box attack iteration over one batch of images:
  EoT repeat (x32) -> NVAE purify (encode + decode) -> VGG-11 classify -> EoT-mean logits -> CE loss
  -> backward-to-input through classifier and purifier -> PGD-Linf sign step + projection,
with fresh N(0,1) latent noise every step, synthetic images already resident in HBM.
Workload at N=1: BASELINE.json configs[1] — NVAE purify, CelebA-64 shapes, bs = 256 images (x EoT 32 = 8192 defender
rows), fp32-class arithmetic, alphas of configs/ours_cosine_no_preprocessing_ids.yaml x 0.7, assumed NVAE config of
SURVEY.md §6.  The 8192 rows of a step run as chunks of --chunk-rows rows (default 1024: 90 GB of activations per chunk plan),
alternating over --streams engines on their own HIP streams so that one chunk's kernel tails overlap the other's.
N>1: the same batch per rank (weak scaling, images are independent), one RCCL all-gather of accuracy counters.
"""
import argparse
import json
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0     # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense"
# precision 'bf16x3': every fp32-class product costs 3 bf16 MFMAs (hi*hi + hi*lo + lo*hi), so the ceiling of the
# ALGORITHMIC flop rate of the dominant kernel is a third of the dense bf16 peak.
PEAK_BF16X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0
PEAK_HBM_TBPS = 8.0                # same guide, "HBM3E peak BW 8.0 TB/s spec" (6.29 TB/s measured by its float4 copy)


def conv_algorithmic_flops(plan):
    """2 * MACs of every ga_conv2d launch in a plan (dense taps; transposed convs counted at their forward size)."""
    from gen_adversarial_amd import _lib as L
    total = 0
    for d in plan.descs:
        if isinstance(d, L.ConvDesc):
            ctot = d.C1 + d.C2
            pix = d.N * d.Ho * d.Wo if d.sd == 1 else d.N * d.Hi * d.Wi
            total += 2 * pix * d.KH * d.KW * ctot * d.Cout
    return total


def dec_cell_algorithmic_flops(plan):
    """2 * MACs of the contractions a fused decoder-cell launch stands for: forward both 1x1 convs, backward the transpose of
    the second one (the recomputed first conv is extra work, not algorithmic; the first one's transpose is its own ga_conv2d)"""
    from gen_adversarial_amd import _lib as L
    return (sum(2 * d.N * d.H * d.W * d.C * d.Hd * (1 if d.backward else 2) for d in plan.descs if isinstance(d, L.DecCellDesc)) +
            # the halo form also carries d x (3 contractions backward) — algorithmic flops only, the halo recompute is not counted
            sum(2 * d.N * d.H * d.W * d.Cin * d.Hd * (3 if d.backward else 2) for d in plan.descs if isinstance(d, L.DecCellHaloDesc)))


def conv_algorithmic_bytes(plan):
    """bytes every ga_conv2d launch of a plan must move at least once: input(s), weights, output, act' source, addends"""
    from gen_adversarial_amd import _lib as L
    total = 0
    for d in plan.descs:
        if isinstance(d, L.ConvDesc):
            pin, pout = d.N * d.Hi * d.Wi, d.N * d.Ho * d.Wo
            total += 4 * (pin * (d.C1 + d.C2) + d.Cout * d.KH * d.KW * (d.C1 + d.C2) + pout * d.Cout)
            total += 4 * pout * d.Cout * (bool(d.dact_x) + bool(d.addend2))
            if d.addend:
                total += 4 * d.Cout * (d.Ho * d.Wo if d.addend_bcast_n else pout // max(1, d.addend_rep))
    return total


def hbm_bound_classes(eng, stream):
    """BASELINE.md §3 / SURVEY.md §8(d): achieved_hbm = algorithmic bytes / kernel time / HBM peak for each memory-bound kernel class
    of one chunk (forward + backward plan): per-op HIP-event durations (ga_plan_profile) and the bytes each op must move at least
    once (its inputs and outputs, fp32; weights and per-row vectors are negligible and not counted)."""
    from gen_adversarial_amd import _lib as L
    acc = {}

    def add(name, nbytes, ms):
        e = acc.setdefault(name, [0.0, 0.0, 0])
        e[0] += nbytes
        e[1] += ms
        e[2] += 1
    for plan in (eng.fwd, eng.bwd):
        for d, ms in zip(plan.descs, plan.profile(stream)):
            if isinstance(d, L.DwDesc):
                pout = d.N * d.H * d.W
                pin = pout // 4 if (d.up2 or d.pool2) else pout
                # forward: read x (low resolution when up2), write y; backward: read dy and the saved input, write dx
                nb = 4 * d.C * ((pin + pout) if not d.dact_x else (pout + pin + (pin if d.pool2 else pout)))
                add('dwconv5 (depthwise 5x5 + SiLU of the unfused decoder cells)', nb, ms)
            elif isinstance(d, L.SeExciteDesc):
                rows = d.N * d.P * d.C * 4
                nb = rows * ((1 + (2 if d.out else 0)) if not d.backward else 2)      # t (+ skip, out) | t, dout
                add('se_excite (squeeze + excite' + (', merge' if d.out else '') + (', backward' if d.backward else '') + ')', nb, ms)
            elif isinstance(d, L.SeApplyDesc):
                add('se_apply (out = skip + 0.1 gate t)', 3 * 4 * d.N * d.H * d.W * d.C, ms)
            elif isinstance(d, L.SamplerDesc):
                px = d.N * d.h * d.w
                nb = 4 * px * ((d.ldq + (d.ldp if d.p else 0) + d.NL + d.ldz) if not d.backward else
                               (d.ldz + d.ldq + (2 * d.ldp if d.p else 0) + d.NL + d.ldq))
                add('sampler (soft clamp, exp, interpolation)' + (' backward' if d.backward else ''), nb, ms)
            elif isinstance(d, L.DmlDesc):
                px = d.N * d.H * d.W
                nb = 4 * px * ((d.ld + 3 + d.ld_img) if not d.backward else (2 * d.ld + d.ld_img + 3))
                add('dml_mean (mixture mean + clamp chain)' + (' backward' if d.backward else ''), nb, ms)
            elif isinstance(d, L.ImageIoDesc):
                add('image_io (EoT repeat, NCHW <-> NHWC)', 4 * d.N * d.H * d.W * (d.ld + 3), ms)
            elif isinstance(d, L.AxpbyDesc):
                add('axpby (gradient accumulation)', 4 * d.n * (3 if d.beta else 2), ms)
            elif isinstance(d, L.MaxpoolDesc):
                add('maxpool2 (VGG)', 4 * d.N * d.H * d.W * d.C * (1.25 if not d.backward else 1.5), ms)
    out = {}
    for name, (nb, ms, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        tbps = nb / (ms / 1e3) / 1e12 if ms > 0 else None
        out[name] = {'launches_per_chunk': n, 'ms_per_chunk': ms, 'algorithmic_mb_per_chunk': nb / 1e6, 'achieved_tbps': tbps,
                     'achieved_hbm': (tbps / PEAK_HBM_TBPS) if tbps else None}
    return out


ROCPROF = '/opt/rocm/bin/rocprofv3'


def pmc_child(args):
    """--pmc-child: what the two counter passes profile — ONE forward + backward replay of the plans of --pmc-workload (the
    launches `roofline.achieved` is about).  Started by measure_pmc_traffic() under `rocprofv3 --pmc <counter> --kernel-trace`."""
    if args.pmc_workload == 'e4e':
        eng, _ = build_e4e_defender('cuda:0', 32, args.eot, args.precision)
        eng.noise.normal_()
        eng.noise_coef.copy_(eng.noise_eps / eng.noise.flatten(1).norm(dim=1))
    elif args.pmc_workload == 'trans':
        eng, _ = build_trans_defender('cuda:0', 64, args.eot, args.precision)
    else:
        eng, _ = build_model('cuda:0', args.chunk_rows, args.eot, seed=0, precision=args.precision, share_encoder=args.share_encoder)
    eng.x_in.uniform_()
    for e in eng.eps:
        e.normal_()
    eng.forward()
    eng.dlogits.normal_()
    eng.backward()
    torch.cuda.synchronize()


def measure_pmc_traffic(args, workload='nvae'):
    """HBM bytes per conv launch, measured in THIS run: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: the TCC has
    4 counter slots, they cost 3 + 2) over a child process that replays one chunk's forward + backward plan.  Units and gfx950
    correction as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: the counters are in KB; FETCH_SIZE reports
    half of the bytes of wide coalesced reads (doubled here), WRITE_SIZE is exact for 16-B-per-lane streaming stores.
    Runs BEFORE this process touches the GPU (children of a GPU-initialised process must not be exec'ed on this pool).
    Returns (bytes per conv launch or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = ROCPROF if os.path.exists(ROCPROF) else shutil.which('rocprofv3')
    if exe is None:
        return None, 'rocprofv3 not found'
    sums = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix=f'ga_pmc_{counter.lower()}_', dir=os.environ.get('TMPDIR', '/tmp'))
        cmd = [exe, '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'pmc', '--',
               sys.executable, os.path.abspath(__file__), '--pmc-child', '--pmc-workload', workload, '--chunk-rows', str(args.chunk_rows),
               '--eot', str(args.eot), '--precision', args.precision] + (['--share-encoder'] if args.share_encoder else [])
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.pmc_timeout, cwd=os.environ.get('TMPDIR', '/tmp'))
        except subprocess.TimeoutExpired:
            shutil.rmtree(d, ignore_errors=True)
            return None, f'{counter} pass timed out after {args.pmc_timeout} s'
        files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
        if r.returncode != 0 or not files:
            shutil.rmtree(d, ignore_errors=True)
            return None, f'{counter} pass failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}'
        tot, n = 0.0, 0
        for f in files:
            with open(f, newline='') as fh:
                for row in csv.DictReader(fh):
                    name = row['Kernel_Name']
                    if row['Counter_Name'] == counter and 'ga::conv_' in name and 'splitk' not in name:
                        tot += float(row['Counter_Value'])
                        n += 1
        shutil.rmtree(d, ignore_errors=True)
        if n == 0:
            return None, f'{counter} pass saw no conv launch'
        sums[counter] = (tot, n)
    fetch, nf = sums['FETCH_SIZE']
    write, nw = sums['WRITE_SIZE']
    per_launch = 2.0 * fetch * 1024.0 / nf + write * 1024.0 / nw
    return per_launch, (f'measured in this run before the timed region: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, '
                        f'--kernel-trace only) over one forward + backward replay of '
                        + {'nvae': f'a {args.chunk_rows}-row chunk plan', 'e4e': 'the 32-row configs[2] plan', 'trans': 'the 64-row configs[4] plan'}[workload]
                        + f' in a child process; '
                        f'{nf} conv launches; bytes = 2 x FETCH_SIZE KB (gfx950 counts half of wide coalesced reads) + WRITE_SIZE KB')


def measured_peaks(device):
    """The two peaks the roofline figures are normalised by, MEASURED on this box with the library's
    microbenchmarks (csrc/microbench.hip), HIP events on the current stream: a 16-B-per-lane copy of 1 GiB (read + write bytes)
    and a bare v_mfma_f32_32x32x16_bf16 loop on pseudo-random operands (2 workgroups of 4 waves per CU)."""
    from gen_adversarial_amd import _lib as L
    st = torch.cuda.current_stream(device).cuda_stream
    n = 256 * 1024 * 1024
    src = torch.empty(n, device=device).normal_()
    dst = torch.empty_like(src)
    blocks, iters = 512, 20000
    out = torch.empty(blocks * 256, device=device)

    def timed(fn, reps):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / 1e3 / reps
    t_copy = timed(lambda: L.check(L.lib.ga_microbench_hbm_copy(src.data_ptr(), dst.data_ptr(), n, st), 'hbm_copy'), 10)
    t_mfma = timed(lambda: L.check(L.lib.ga_microbench_mfma_bf16(out.data_ptr(), blocks, iters, st), 'mfma_bf16'), 5)
    t_mfma16 = timed(lambda: L.check(L.lib.ga_microbench_mfma_bf16_shape(out.data_ptr(), blocks, iters, 16, st), 'mfma_bf16_16'), 5)
    flops = blocks * 4 * iters * 8 * 2 * 32 * 32 * 16
    del src, dst, out
    return {'hbm_copy_tbps': 2 * 4 * n / t_copy / 1e12, 'bf16_mfma_tflops': flops / t_mfma / 1e12,
            # the other MFMA shape (the kernels use 32x32x16: the ceiling above stays the one they are normalised by).  Under the chip's
            # power limit the 16x16x32 loop holds a higher clock; DESIGN.md §7 has what that is worth inside the conv kernel (2 - 7 %)
            'bf16_mfma_tflops_16x16x32': flops / t_mfma16 / 1e12,
            'how': 'ga_microbench_hbm_copy (1 GiB float4 copy, read + write bytes / time) and ga_microbench_mfma_bf16 (bare '
                   '32x32x16 bf16 MFMA loop, pseudo-random operands, 512 workgroups x 4 waves x 160000 MFMAs), HIP events, this box'}


def build_model(device, rows, rep, seed=0, precision='bf16x3', share_encoder=False, store=None):
    from gen_adversarial_amd.engine import Engine
    from gen_adversarial_amd.nvae_spec import ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, init_nvae_state_dict
    from gen_adversarial_amd.vgg_spec import build_vgg_spec, init_vgg_state_dict
    with open(os.path.join(ROOT, 'configs', 'ours_cosine_no_preprocessing_ids.yaml')) as f:
        y = yaml.safe_load(f)
    alphas = [a * y['alpha_attenuation'] for a in y['interpolation_alphas']]
    sd = init_nvae_state_dict(ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, seed)
    vspec = build_vgg_spec(100, 1)
    vsd = init_vgg_state_dict(100, 1, seed + 1)
    eng = Engine(sd, ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, vsd, vspec, rows=rows, rep=rep, alphas=alphas,
                 temperature=0.6, noise_eps=float(y['initial_noise_eps']), device=device, precision=precision,
                 share_encoder=share_encoder, store=store)
    return eng, (sd, vsd, vspec, alphas)


class AttackStep:
    """PGD-Linf iteration (eps 8/255, step 2/255) on `images`, processed as chunks of eng.rows // rep images that
    alternate over the engines (one HIP stream each; engines share the folded weights)."""

    def __init__(self, engines, streams, labels, x_orig, eps=8.0 / 255.0, step=2.0 / 255.0, bpda=False):
        self.engines, self.streams = engines, streams
        self.bpda = bpda          # BPDA (the attack BASELINE.json configs[3] names): backward through the classifier only
        self.labels, self.x_orig, self.eps, self.step_size = labels, x_orig, eps, step
        self.x_adv = x_orig.clone()
        self.rep = engines[0].rep
        self.per = engines[0].rows // self.rep                    # images per chunk
        assert x_orig.shape[0] % self.per == 0
        self.logits = torch.zeros(x_orig.shape[0], engines[0].logits.shape[-1], device=x_orig.device)

    def chunk(self, eng, lo, hi):
        eng.x_in.copy_(self.x_adv[lo:hi])
        for e in eng.eps:
            e.normal_()
        eng.forward()
        logits = eng.logits.view(-1, self.rep, eng.logits.shape[-1]).mean(dim=1)          # EoT mean
        p = torch.softmax(logits, dim=1)
        p[torch.arange(p.shape[0], device=p.device), self.labels[lo:hi]] -= 1.0           # d CE / d mean-logits
        eng.dlogits.view(-1, self.rep, p.shape[-1]).copy_((p / self.rep).unsqueeze(1).expand(-1, self.rep, -1))
        eng.backward(identity_purifier=True) if self.bpda else eng.backward()
        nxt = self.x_adv[lo:hi] + self.step_size * eng.dx.sign()
        xo = self.x_orig[lo:hi]
        self.x_adv[lo:hi] = torch.min(torch.max(nxt, xo - self.eps), xo + self.eps).clamp_(0.0, 1.0)
        self.logits[lo:hi] = logits

    def forward_only(self):
        """clean EoT-mean logits of x_adv (used once to pick the labels)"""
        self._run(lambda eng, lo, hi: self._fwd(eng, lo, hi))
        return self.logits

    def _fwd(self, eng, lo, hi):
        eng.x_in.copy_(self.x_adv[lo:hi])
        for e in eng.eps:
            e.normal_()
        eng.forward()
        self.logits[lo:hi] = eng.logits.view(-1, self.rep, eng.logits.shape[-1]).mean(dim=1)

    def _run(self, fn):
        if self.streams[0] is None:                          # stub rehearsal on the CPU: no streams
            for c in range(self.x_orig.shape[0] // self.per):
                fn(self.engines[c % len(self.engines)], c * self.per, (c + 1) * self.per)
            return
        main = torch.cuda.current_stream()
        for s in self.streams:
            s.wait_stream(main)
        for c in range(self.x_orig.shape[0] // self.per):
            k = c % len(self.engines)
            with torch.cuda.stream(self.streams[k]):
                fn(self.engines[k], c * self.per, (c + 1) * self.per)
        for s in self.streams:
            main.wait_stream(s)

    def __call__(self):
        self._run(self.chunk)
        return self.logits


def Engine_clone(eng, model, device, args):
    """clone_engine with the source engine's share_encoder setting"""
    from gen_adversarial_amd.engine import Engine
    from gen_adversarial_amd.nvae_spec import ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION
    sd, vsd, vspec, alphas = model
    return Engine(sd, ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, vsd, vspec, rows=eng.rows, rep=eng.rep, alphas=alphas,
                  temperature=0.6, noise_eps=eng.noise_eps, device=device, precision=args.precision,
                  share_encoder=eng.share_encoder, store=eng.store)


def clone_engine(eng, model, device, args, precision=None):
    """a second engine over the same folded weights (WeightStore) with its own activations, for another stream"""
    from gen_adversarial_amd.engine import Engine
    from gen_adversarial_amd.nvae_spec import ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION
    sd, vsd, vspec, alphas = model
    return Engine(sd, ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, vsd, vspec, rows=eng.rows, rep=eng.rep, alphas=alphas,
                  temperature=0.6, noise_eps=eng.noise_eps, device=device, precision=precision or args.precision,
                  share_encoder=args.share_encoder, store=eng.store)


def cpu_baseline(model, rows, rep, check=None):
    """the oracle (CPU restatement) on the host cores: one attack step (forward + input-gradient) on `rows` rows.
    `check(x, eps, logits, grad)`, when given, receives the oracle's inputs and results after the timed part (the
    caller replays them on the HIP path: a full-size parity figure next to the timing)."""
    from oracle import defender_oracle as D
    from gen_adversarial_amd.nvae_spec import ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, build_spec
    sd, vsd, vspec, alphas = model
    spec = build_spec(ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # a one-GPU box gives this process a 16-CPU share whatever the host's core count (oversubscribing stalls OpenMP)
    threads = int(os.environ.get('GA_CPU_THREADS', min(cores, 16)))
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(rows // rep, 3, 64, 64, generator=g).requires_grad_(True)
    eps = [torch.randn(rows, spec.num_latent, gs.res, gs.res, generator=g) for gs in spec.groups]
    noise = torch.randn(rows, 3, 64, 64, generator=g)
    t0 = time.time()
    logits, _ = D.nvae_defender(sd, spec, vsd, vspec, x.repeat_interleave(rep, dim=0), alphas, eps, noise, 0.0)
    mean = logits.view(-1, rep, logits.shape[-1]).mean(dim=1)
    loss = torch.nn.functional.cross_entropy(mean, mean.argmax(dim=1).detach(), reduction='sum')
    (gx,) = torch.autograd.grad(loss, [x])
    dt = time.time() - t0
    if check is not None:
        check(x.detach(), eps, logits.detach(), gx)
    return {'value': rows / dt, 'unit': 'rows/s', 'cores': threads, 'kind': 'port',
            'sample': f'{rows // rep} image(s) x EoT {rep} = {rows} rows, one attack step (forward + input-gradient), '
                      f'{dt:.1f} s of oracle (PyTorch CPU fp32) time, dX only'}


def free_gpu_memory():
    """engines hold reference cycles (activation records <-> engine): collect them before returning their HBM"""
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def _time_steps(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def secondary_measurements(out, args, device, model, store, x, labels):
    """Driver-visible figures beside the headline (never `value`): the exact-fp32 arithmetic at the same configuration, the
    reference's own protocol (ONE image x EoT 32 per defender call: src/experiments/test_defense.py:116,60) eager and as HIP
    graphs, and BASELINE.json configs[2] (e4e + StyleGAN2-1024 + ResNet-50, 256 px, eps 4.0) with its own roofline block."""
    sec = out.setdefault('secondary', {})
    # ---- exact fp32 (v_mfma_f32_32x32x2_f32) at the headline configuration: the same images, chunks and streams as `value`
    try:
        log('secondary: fp32 precision at the headline configuration ...')
        e32, m32 = build_model(device, args.chunk_rows, args.eot, seed=0, precision='fp32', share_encoder=False)
        n_chunks = args.images * args.eot // args.chunk_rows
        n_eng = max(1, min(args.streams, n_chunks))
        engs = [e32] + [clone_engine(e32, m32, device, args, precision='fp32') for _ in range(n_eng - 1)]
        st = AttackStep(engs, [torch.cuda.Stream(device=device) for _ in engs], labels.clone(), x.clone())
        t = _time_steps(st, 2, warm=1)
        f_ms, fc_ms, fn = e32.fwd.time(e32.stream(), iters=1, per_conv=True)
        b_ms, bc_ms, bn = e32.bwd.time(e32.stream(), iters=1, per_conv=True)
        fl = conv_algorithmic_flops(e32.fwd) + conv_algorithmic_flops(e32.bwd)
        sec['fp32_precision'] = {'rows_per_s': args.images * args.eot / t, 'ms_per_step': t * 1e3, 'dtype': 'f32 (exact f32 MFMA)',
                                 'what': f'{args.images} images x EoT {args.eot} per step ({args.chunk_rows}-row chunks, {n_eng} streams): the headline '
                                         'workload, images, chunking and streams with every contraction on v_mfma_f32_32x32x2_f32 (no fused decoder '
                                         'cell and no tile 8 in this mode: their resident operands are split-bf16 fragments)',
                                 'roofline': {'bound': 'mfma', 'achieved': fl / ((fc_ms + bc_ms) / 1e3) / 1e12, 'peak': PEAK_FP32_MFMA_TFLOPS,
                                              'unit': 'TFLOP/s', 'frac': fl / ((fc_ms + bc_ms) / 1e3) / 1e12 / PEAK_FP32_MFMA_TFLOPS}}
        del st, e32, engs
        free_gpu_memory()
    except Exception as ex:
        sec['fp32_precision'] = {'rows_per_s': None, 'what': f'failed: {ex}'}
    # ---- the reference protocol: one image x EoT 32 per call
    try:
        log('secondary: reference protocol (1 image x EoT 32) ...')
        e1, _ = build_model(device, args.eot, args.eot, seed=0, precision=args.precision, share_encoder=False, store=store)
        st = AttackStep([e1], [torch.cuda.Stream(device=device)], labels[:1].clone(), x[:1].clone())
        t_eager = _time_steps(st, 8, warm=2)
        with torch.cuda.stream(st.streams[0]):
            e1.enable_graphs()
        t_graph = _time_steps(st, 8, warm=2)
        e1.disable_graphs()
        sec['reference_protocol_1_image'] = {'rows_per_s_eager': args.eot / t_eager, 'ms_per_step_eager': t_eager * 1e3,
                                             'rows_per_s_hip_graph': args.eot / t_graph, 'ms_per_step_hip_graph': t_graph * 1e3,
                                             'launches_per_step': len(e1.fwd) + len(e1.bwd),
                                             'what': f'1 image x EoT {args.eot} = {args.eot} defender rows per attack step (the batch size the reference\'s '
                                                     'attacks use); literal x.repeat(eot) path'}
        del st, e1
        free_gpu_memory()
        # the same call through the defender API's default: configs[1]'s yaml has initial_noise_eps 0.0, so the 32 EoT replicas of the
        # image are identical up to the first latent draw and the encoder runs once per image (exact) — what a drop-in
        # user of the reference's protocol gets
        e1s, _ = build_model(device, args.eot, args.eot, seed=0, precision=args.precision, share_encoder=True, store=store)
        sts = AttackStep([e1s], [torch.cuda.Stream(device=device)], labels[:1].clone(), x[:1].clone())
        t_sh = _time_steps(sts, 8, warm=2)
        sec['reference_protocol_1_image'].update({
            'rows_per_s_api_default_shared_encoder': args.eot / t_sh, 'ms_per_step_api_default_shared_encoder': t_sh * 1e3,
            'launches_per_step_api_default_shared_encoder': len(e1s.fwd) + len(e1s.bwd),
            'api_default_note': 'NVAEDefenseModel runs the (deterministic) encoder once per image when no input noise is configured '
                                '(configs[1]: initial_noise_eps 0.0); same logits and gradients as the literal repeat'})
        del sts, e1s
        free_gpu_memory()
    except Exception as ex:
        sec.setdefault('reference_protocol_1_image', {})['failure'] = f'failed: {ex}'
        sec['reference_protocol_1_image'].setdefault('rows_per_s_eager', None)
    # ---- per-class input gradients (DeepFool: 10 classes; FAB on the ids experiment: 100) — one forward + one backward per class
    #      (the reference: src/attacks/untargeted.py:526-560, :605-635) against the K-cotangent backward plan
    try:
        log('secondary: per-class gradients, K-cotangent plan ...')
        sec['class_jacobian'] = class_jacobian_measurement(args, device, store)
    except Exception as ex:
        sec['class_jacobian'] = {'what': f'failed: {type(ex).__name__}: {ex}'}
    free_gpu_memory()
    # ---- configs[2]: e4e + StyleGAN2-1024 defender, ResNet-50, 256 px, 64 rows per step, input noise eps 4.0
    try:
        log('secondary: configs[2] e4e + StyleGAN2 defender ...')
        sec['configs2_e4e_defender'] = e4e_defender_measurement(args, device)
    except Exception as ex:
        sec['configs2_e4e_defender'] = {'rows_per_s': None, 'what': f'failed: {type(ex).__name__}: {ex}'}
    free_gpu_memory()
    # ---- configs[4]: Style-Transformer + StyleGAN2-512 defender, ResNeXt-50, 128 px
    try:
        log('secondary: configs[4] Style-Transformer defender ...')
        sec['configs4_trans_defender'] = trans_defender_measurement(args, device)
    except Exception as ex:
        sec['configs4_trans_defender'] = {'rows_per_s': None, 'what': f'failed: {type(ex).__name__}: {ex}'}
    free_gpu_memory()


def class_jacobian_measurement(args, device, store):
    """What one DeepFool iteration (10 class gradients) and one FAB gradient evaluation on the ids defender (100 classes) cost:
    ONE forward + C backward replays of the plain plan (the reference's per-class `.backward(retain_graph=True)` loop) against ONE
    forward + ceil(C / K) replays of the K-cotangent plan, for the reference's protocol (1 image x EoT 32,
    K = 16) and a 16-image batch (512 rows, K = 4).  API default: encoder shared by the EoT replicas (no input noise)."""
    import math
    from gen_adversarial_amd.engine import Engine
    from gen_adversarial_amd.nvae_spec import ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION
    res = {}
    for images, K in (((1, 16), (16, 4)) if args.class_jacobian_batch else ((1, 16),)):
        rows = images * args.eot
        e1, model = build_model(device, rows, args.eot, seed=0, precision=args.precision, share_encoder=True, store=store)
        sd, vsd, vspec, alphas = model
        eK = Engine(sd, ASSUMED_NVAE_CONFIG, ASSUMED_NVAE_RESOLUTION, vsd, vspec, rows=rows, rep=args.eot, alphas=alphas, temperature=0.6,
                    noise_eps=0.0, device=device, precision=args.precision, share_encoder=True, store=store, cot_rep=K)
        for e in (e1, eK):
            e.x_in.uniform_()
            for b in e.eps:
                b.normal_()
            e.dlogits.normal_()

        def iteration(e, C):
            e.forward()
            for _ in range(math.ceil(C / e.cot_rep)):
                e.backward()
        entry = {'rows': rows, 'K': K, 'what': f'{images} image(s) x EoT {args.eot}: one forward + the input gradients of C class logits'}
        for C, name in ((10, 'deepfool_10_classes'), (100, 'fab_100_classes')):
            n = 2 if C * rows > 2000 else 5
            t1 = _time_steps(lambda: iteration(e1, C), n, warm=1)
            tK = _time_steps(lambda: iteration(eK, C), n, warm=1)
            entry[name] = {'ms_per_class_loop': t1 * 1e3, 'backward_replays_per_class_loop': C,
                           'ms_k_cotangent': tK * 1e3, 'backward_replays_k_cotangent': math.ceil(C / K), 'speedup': t1 / tK}
        res[f'{images}_images'] = entry
        del e1, eK
        free_gpu_memory()
    return res


def defender_parity(eng, oracle_call, res_px, n_latent, noise_eps, eps_std=1.0):
    """parity_vs_oracle of a configs[2] / configs[4] plan AS TIMED: row 0 of the plan (image 0 under seeded draws) against the CPU
    oracle evaluated on that one row, forward only (tests/test_fullsize_configs_gpu.py holds the gradient comparison)."""
    if getattr(eng, 'dry_run', False):
        return None
    gen = torch.Generator().manual_seed(77)
    x = torch.rand(eng.x_in.shape[0], 3, res_px, res_px, generator=gen)
    z = eps_std * torch.randn(eng.rows, n_latent, 512, generator=gen)
    eng.x_in.copy_(x.to(eng.x_in.device))
    eng.eps[0].copy_(z.to(eng.x_in.device))
    nz = None
    if noise_eps:
        nz = torch.randn(eng.rows, 3, res_px, res_px, generator=gen)
        eng.noise.copy_(nz.to(eng.x_in.device))
        eng.noise_coef.copy_((noise_eps / nz.flatten(1).norm(dim=1)).to(eng.x_in.device))
    eng.forward()
    torch.cuda.synchronize()
    torch.set_num_threads(int(os.environ.get('GA_CPU_THREADS', min(len(os.sched_getaffinity(0)), 16))))   # the one-GPU box's CPU share
    log(f'   parity: oracle on row 0 ({torch.get_num_threads()} threads) ...')
    with torch.no_grad():
        lg, pur = oracle_call(x[:1], z[:1], None if nz is None else nz[:1])
    return {'rows': 1, 'max_abs_logit_err': float((eng.logits.view(eng.rows, -1)[:1].cpu() - lg).abs().max()),
            'max_abs_logit': float(lg.abs().max()), 'max_abs_purified_err': float((eng.purified_nchw()[:1].cpu() - pur).abs().max()),
            'note': 'row 0 of the timed plan vs the CPU oracle on that row (forward); tolerance of the path 1e-3 (relative to max |logit| '
                    'for logits); gradients: tests/test_fullsize_configs_gpu.py'}


def _scale_head(csd, s):
    """The random-weight classifiers behind the random-weight generators put out logits of |20| (ResNeXt-50) to |160| (ResNet-50).
    The split-bf16 path carries a RELATIVE error of ~1.7e-5 of max |logit| through these networks (measured: 2.7e-3 at |163|, 1.7e-4
    at |10|; the CPU oracle's own fp32-vs-fp64 difference on the classifier is 4e-6 at |10|, so this is the arithmetic, not the
    comparison), which meets north_star's ABSOLUTE 1e-3 for |logits| up to ~60.  Trained classifiers put out |logits| of 10 - 20:
    the head's last layer is scaled so that the random-weight stand-ins do too, and tests/test_fullsize_configs_gpu.py asserts the
    absolute bar there (a bound relative to |83| would let a 50x regression pass).  No effect on the arithmetic
    that is timed."""
    csd['model.fc.3.weight'] = csd['model.fc.3.weight'] * s
    csd['model.fc.3.bias'] = csd['model.fc.3.bias'] * s


def build_trans_defender(device, rows, eot, precision, parts=False):
    """BASELINE.json configs[4] (configs/ours_learned_blur_cars.yaml: 16 learned alphas x 0.7, Gaussian blur of the input): the
    Style-Transformer encoder (IR-SE50 at 192 x 256 + 3 decoder layers over 16 queries) + StyleGAN2-512 + ResNeXt-50 32x4d at
    128 px, random weights of the reference architecture"""
    from gen_adversarial_amd.engine import Engine
    from gen_adversarial_amd.trans_spec import build_trans_spec, init_trans_state_dict
    from gen_adversarial_amd.resnet_spec import build_resnet_spec, init_resnet_state_dict
    from gen_adversarial_amd.stylegan_spec import build_stylegan_spec, init_stylegan_state_dict
    with open(os.path.join(ROOT, 'configs', 'ours_learned_blur_cars.yaml')) as f:
        y = yaml.safe_load(f)
    alphas = [a * y['alpha_attenuation'] for a in y['interpolation_alphas']]
    tspec, tsd = build_trans_spec(1), init_trans_state_dict(1, 0)
    gspec = build_stylegan_spec(512)
    gsd = init_stylegan_state_dict(gspec, 1)
    cspec, csd = build_resnet_spec(4, 1, (3, 4, 6, 3), 32, 4), init_resnet_state_dict(4, 1, 2, (3, 4, 6, 3), 32, 4)
    _scale_head(csd, 0.5)
    avg = 0.1 * torch.randn(16, 512, generator=torch.Generator().manual_seed(5))
    eng = Engine.bare(rows, device=device, precision=precision, rep=eot, resolution=(3, 128, 128), alphas=alphas,
                      noise_eps=float(y['initial_noise_eps']), blur=bool(y['gaussian_blur_input']), share_encoder=False)
    eng.build_trans_defense(tsd, tspec, gsd, gspec, avg, csd, cspec, pool_to=128)
    if parts:       # tests/test_fullsize_configs_gpu.py: the same weights for the CPU oracle
        return eng, y, (tsd, tspec, gsd, gspec, avg, csd, cspec, alphas)
    return eng, y


def trans_defender_measurement(args, device, rows=64, eot=32):
    """configs[4] on one GPU: one PGD-Linf iteration (forward + input gradient) over 2 images x EoT 32 = 64 defender rows in one
    plan run, literal x.repeat(eot) path.  (The config's "bf16" is the reference's autocast setting; arithmetic here is the
    engine's fp32-class split-bf16 mode.)"""
    eng, y, parts = build_trans_defender(device, rows, eot, args.precision, parts=True)
    pmc = getattr(args, 'defender_pmc', {}).get('trans', (None, 'PMC passes skipped'))
    n_img = rows // eot
    g = torch.Generator(device=device).manual_seed(9)
    x = torch.rand(n_img, 3, 128, 128, device=device, generator=g)
    x_adv = x.clone()
    labels = torch.zeros(n_img, dtype=torch.long, device=device)

    def step():
        eng.x_in.copy_(x_adv)
        eng.eps[0].normal_().mul_(eng.eps_std)
        eng.forward()
        lg = eng.logits.view(-1, eot, eng.logits.shape[-1]).mean(dim=1)
        p = torch.softmax(lg, dim=1)
        p[torch.arange(p.shape[0], device=p.device), labels] -= 1.0
        eng.dlogits.view(-1, eot, p.shape[-1]).copy_((p / eot).unsqueeze(1).expand(-1, eot, -1))
        eng.backward()
        nxt = x_adv + (2.0 / 255.0) * eng.dx.sign()
        x_adv.copy_(torch.min(torch.max(nxt, x - 8.0 / 255.0), x + 8.0 / 255.0).clamp_(0.0, 1.0))
    t = _time_steps(step, 5, warm=2)
    s = eng.stream()
    f_ms, fc_ms, fn = eng.fwd.time(s, iters=1, per_conv=True)
    b_ms, bc_ms, bn = eng.bwd.time(s, iters=1, per_conv=True)
    flops = conv_algorithmic_flops(eng.fwd) + conv_algorithmic_flops(eng.bwd)
    achieved = flops / ((fc_ms + bc_ms) / 1e3) / 1e12
    peak = PEAK_BF16X3_TFLOPS if args.precision == 'bf16x3' else PEAK_FP32_MFMA_TFLOPS
    res = {'rows_per_s': rows / t, 'ms_per_step': t * 1e3, 'rows_per_step': rows,
           'what': f'configs[4]: blur -> resize 256 / crop -> Style-Transformer encoder (IR-SE50 @192x256 + 3 decoder layers) -> StyleGAN2-512 -> '
                   f'pool / band / resize 128 -> ResNeXt-50 32x4d, {n_img} images x EoT {eot} = {rows} defender rows per PGD step (forward + input '
                   f'gradient), alphas ours_learned_blur_cars.yaml, {len(eng.fwd)} + {len(eng.bwd)} launches per plan, '
                   f'{eng.bytes / 1e9:.0f} GB of activations + weights',
           'roofline': {'bound': 'mfma', 'kernel': 'ga::conv_* (implicit-GEMM conv family)', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                        'frac': achieved / peak, 'traffic': pmc[0], 'traffic_note': pmc[1],
                        'algorithmic_bytes_per_launch': (conv_algorithmic_bytes(eng.fwd) + conv_algorithmic_bytes(eng.bwd)) / (fn + bn),
                        'launches_per_plan': int(fn + bn), 'avg_launch_ms': (fc_ms + bc_ms) / (fn + bn),
                        'algorithmic_gflop_per_plan': flops / 1e9, 'conv_ms_per_plan': fc_ms + bc_ms, 'plan_ms_fwd': f_ms, 'plan_ms_bwd': b_ms}}
    if not args.no_cpu_baseline and args.defender_parity:
        from oracle import defender_oracle as D, trans_oracle as T          # the checker, never the thing measured
        tsd, tspec, gsd, gspec, avg, csd, cspec, alphas = parts

        def call(x1, z1, _):
            p = T.trans_purify(tsd, tspec, gsd, gspec, avg, D.apply_gaussian_blur(x1), alphas, z1)
            return D.resnet_classifier_call(csd, cspec, p), p
        res['parity_vs_oracle'] = defender_parity(eng, call, 128, 16, 0.0, eps_std=0.8)
    del eng, parts
    return res


def build_e4e_defender(device, rows, eot, precision, parts=False):
    """BASELINE.json configs[2] (configs/ours_cosine_noise_gender.yaml: 18 cosine alphas, initial_noise_eps 4.0): IR-SE50 e4e
    encoder on 256 px -> 18 x 512 latents mixed with mapped noise -> StyleGAN2 at 1024 px -> face_pool 256 -> ResNet-50, random
    weights of the reference architecture"""
    from gen_adversarial_amd.engine import Engine
    from gen_adversarial_amd.e4e_spec import build_e4e_spec, init_e4e_state_dict
    from gen_adversarial_amd.resnet_spec import build_resnet_spec, init_resnet_state_dict
    from gen_adversarial_amd.stylegan_spec import build_stylegan_spec, init_stylegan_state_dict
    with open(os.path.join(ROOT, 'configs', 'ours_cosine_noise_gender.yaml')) as f:
        y = yaml.safe_load(f)
    alphas = [a * y['alpha_attenuation'] for a in y['interpolation_alphas']]
    espec, esd = build_e4e_spec(1024), init_e4e_state_dict(1024, 1, 0)
    gspec = build_stylegan_spec(1024)
    gsd = init_stylegan_state_dict(gspec, 1)
    cspec, csd = build_resnet_spec(2), init_resnet_state_dict(2, 1, 2)
    _scale_head(csd, 1.0 / 16.0)
    avg = 0.1 * torch.randn(18, 512, generator=torch.Generator().manual_seed(5))
    eng = Engine.bare(rows, device=device, precision=precision, rep=eot, resolution=(3, 256, 256), alphas=alphas,
                      noise_eps=float(y['initial_noise_eps']))
    eng.build_e4e_defense(esd, espec, gsd, gspec, avg, csd, cspec, pool_to=256)
    if parts:       # tests/test_fullsize_configs_gpu.py: the same weights for the CPU oracle
        return eng, y, (esd, espec, gsd, gspec, avg, csd, cspec, alphas)
    return eng, y


def e4e_defender_measurement(args, device, rows=64, chunk=32, eot=32):
    """configs[2] on one GPU: forward + backward-to-input (one PGD-Linf iteration), 64 defender rows per step as two 32-row plan
    runs (1 image x EoT 32 each; a 32-row plan holds ~100 GB of activations).  Random weights, synthetic images."""
    eng, y, parts = build_e4e_defender(device, chunk, eot, args.precision, parts=True)
    pmc = getattr(args, 'defender_pmc', {}).get('e4e', (None, 'PMC passes skipped'))
    n_img = rows // eot
    g = torch.Generator(device=device).manual_seed(7)
    x = torch.rand(n_img, 3, 256, 256, device=device, generator=g)
    x_adv = x.clone()
    labels = torch.zeros(n_img, dtype=torch.long, device=device)
    per = chunk // eot

    def step():
        for c in range(n_img // per):
            lo, hi = c * per, (c + 1) * per
            eng.x_in.copy_(x_adv[lo:hi])
            eng.eps[0].normal_()
            eng.noise.normal_()
            eng.noise_coef.copy_(eng.noise_eps / eng.noise.flatten(1).norm(dim=1))
            eng.forward()
            lg = eng.logits.view(-1, eot, eng.logits.shape[-1]).mean(dim=1)
            p = torch.softmax(lg, dim=1)
            p[torch.arange(p.shape[0], device=p.device), labels[lo:hi]] -= 1.0
            eng.dlogits.view(-1, eot, p.shape[-1]).copy_((p / eot).unsqueeze(1).expand(-1, eot, -1))
            eng.backward()
            nxt = x_adv[lo:hi] + (2.0 / 255.0) * eng.dx.sign()
            x_adv[lo:hi] = torch.min(torch.max(nxt, x[lo:hi] - 8.0 / 255.0), x[lo:hi] + 8.0 / 255.0).clamp_(0.0, 1.0)
    t = _time_steps(step, 3, warm=1)
    s = eng.stream()
    f_ms, fc_ms, fn = eng.fwd.time(s, iters=1, per_conv=True)
    b_ms, bc_ms, bn = eng.bwd.time(s, iters=1, per_conv=True)
    flops = conv_algorithmic_flops(eng.fwd) + conv_algorithmic_flops(eng.bwd)
    achieved = flops / ((fc_ms + bc_ms) / 1e3) / 1e12
    peak = PEAK_BF16X3_TFLOPS if args.precision == 'bf16x3' else PEAK_FP32_MFMA_TFLOPS
    res = {'rows_per_s': rows / t, 'ms_per_step': t * 1e3, 'rows_per_step': rows, 'chunk_rows': chunk,
           'what': f'configs[2]: e4e (IR-SE50 @256 px) + StyleGAN2-1024 + face_pool + ResNet-50, {n_img} images x EoT {eot} = {rows} defender rows per '
                   f'PGD step (forward + input gradient), initial_noise_eps {y["initial_noise_eps"]}, alphas ours_cosine_noise_gender.yaml, '
                   f'{len(eng.fwd)} + {len(eng.bwd)} launches per {chunk}-row plan, {eng.bytes / 1e9:.0f} GB of activations + weights',
           'roofline': {'bound': 'mfma', 'kernel': 'ga::conv_* (implicit-GEMM conv family)', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                        'frac': achieved / peak, 'traffic': pmc[0], 'traffic_note': pmc[1],
                        'algorithmic_bytes_per_launch': (conv_algorithmic_bytes(eng.fwd) + conv_algorithmic_bytes(eng.bwd)) / (fn + bn),
                        'launches_per_chunk': int(fn + bn), 'avg_launch_ms': (fc_ms + bc_ms) / (fn + bn),
                        'algorithmic_gflop_per_chunk': flops / 1e9, 'conv_ms_per_chunk': fc_ms + bc_ms, 'plan_ms_fwd': f_ms, 'plan_ms_bwd': b_ms}}
    if not args.no_cpu_baseline and args.defender_parity:
        from oracle import defender_oracle as D                              # the checker, never the thing measured
        esd, espec, gsd, gspec, avg, csd, cspec, alphas = parts

        def call(x1, z1, n1):
            return D.e4e_defender_call(esd, espec, gsd, gspec, avg, csd, cspec, D.add_gaussian_noise(x1, n1, eng.noise_eps), alphas, z1, 256)
        res['parity_vs_oracle'] = defender_parity(eng, call, 256, 18, eng.noise_eps)
    del eng, parts
    return res


def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


class StubEngine:
    """TEST REHEARSAL ONLY (--stub-engine, gloo): a CPU stand-in with the engine's buffers so that the launcher, the
    rendezvous, the barrier + max-over-ranks timing, the all-gather and the JSON line can be exercised without a GPU.
    It computes nothing of the path; the line it produces says so (`data: stub`)."""

    def __init__(self, rows, rep, n_classes=100):
        self.rows, self.rep, self.noise_eps, self.share_encoder, self.bytes = rows, rep, 0.0, False, 0
        self.x_in = torch.zeros(rows // rep, 3, 64, 64)
        self.eps = [torch.zeros(rows, 4)]
        self.logits = torch.zeros(rows, n_classes)
        self.dlogits = torch.zeros(rows, n_classes)
        self.dx = torch.zeros(rows // rep, 3, 64, 64)
        self.fwd, self.bwd = [], []

    def forward(self):
        self.logits.copy_(self.x_in.mean(dim=(1, 2, 3)).repeat_interleave(self.rep).unsqueeze(1) * torch.arange(1, self.logits.shape[1] + 1))

    def backward(self):
        self.dx.fill_(self.dlogits.sum().item())


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh processes, one rank per GPU, as the reference spawns one
    process per GPU itself (src/experiments/test_defense.py:296-302).  Runs BEFORE anything in this process touches the GPU;
    the children are ordinary `python bench.py` processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what
    torch.distributed.run would set).  Rank 0's stdout (the one JSON line) is passed through; any failing rank fails the run."""
    import socket
    import subprocess
    n = args.gpus
    # nothing here touches the HIP runtime (not even a device count): every child checks its own device and fails loudly
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   LOCAL_WORLD_SIZE=str(n))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE))
    # every rank's stdout is piped (drained concurrently: a full pipe must not block a rank inside a collective); rank 0's is the
    # JSON line, the others' last words are shown when the run fails
    import threading
    outs = [b''] * n

    def drain(i):
        outs[i] = procs[i].stdout.read()
    threads = [threading.Thread(target=drain, args=(i,)) for i in range(n)]
    for t in threads:
        t.start()
    codes = [p.wait() for p in procs]
    for t in threads:
        t.join()
    sys.stdout.write(outs[0].decode())
    sys.stdout.flush()
    if any(codes):
        for r in range(1, n):
            tail = outs[r].decode(errors='replace')[-2000:]
            if codes[r] or tail.strip():
                print(f'---- rank {r} (exit code {codes[r]}) stdout tail ----\n{tail}', file=sys.stderr, flush=True)
        raise SystemExit(f'bench.py --gpus {n}: rank exit codes {codes}')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--images', type=int, default=256, help='images per GPU per step (bs of BASELINE.json configs[1])')
    ap.add_argument('--eot', type=int, default=32)
    ap.add_argument('--chunk-rows', type=int, default=1024,
                    help='defender rows (images x EoT) one plan run processes: a 1024-row chunk holds 90 GB of activations, two of them '
                         '(--streams 2) 180 of the 288 GB.  512-row chunks (95 GB in all) run 1.5 - 1.9 %% slower (interleaved runs on '
                         'one box, gpurun_out/r04_chunk_ab.log): the 3x3 layers at 8x8 and 4x4 then launch ONE wave of workgroups, '
                         'whose setup and epilogue phases coincide on every CU')
    ap.add_argument('--streams', type=int, default=2, help='engines / HIP streams the chunks alternate over')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-rows256', action='store_true', help='skip the secondary 256-row single-plan measurement')
    ap.add_argument('--no-shared-variant', action='store_true', help='skip the secondary shared-encoder measurement')
    ap.add_argument('--share-encoder', action='store_true',
                    help='run the (deterministic) encoder once per image instead of once per EoT replica; identical '
                         'results when initial_noise_eps == 0.  Off by default: the headline number is the literal path')
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl',
                    help='collective backend; gloo (CPU tensors) only to rehearse N>1 on a box with fewer GPUs than ranks')
    ap.add_argument('--precision', choices=['bf16x3', 'fp32'], default='bf16x3',
                    help="dense contractions: 'bf16x3' = 3 bf16 MFMAs per product (logits within ~2e-5 of fp32), 'fp32' = exact f32 MFMA")
    ap.add_argument('--pmc-child', action='store_true', help='internal: the process the PMC passes profile (see measure_pmc_traffic)')
    ap.add_argument('--pmc-workload', choices=['nvae', 'e4e', 'trans'], default='nvae', help='internal: which plans --pmc-child replays')
    ap.add_argument('--class-jacobian-batch', action='store_true',
                    help='secondary.class_jacobian: also time the 16-image batch (512 rows, K = 4), about a minute more')
    ap.add_argument('--defender-pmc', action='store_true',
                    help='also run the PMC passes over the configs[2] / configs[4] defender plans (about 80 s more; off by default so that '
                         'the default run stays inside 4 minutes: profiles/ holds the figures of a run with it)')
    ap.add_argument('--defender-parity', action='store_true',
                    help='secondary configs[2] / configs[4]: also run the CPU oracle on row 0 of the timed plans (about 40 s more; the same '
                         'comparison, on more rows and with gradients, is tests/test_fullsize_configs_gpu.py)')
    ap.add_argument('--no-pmc', action='store_true', help='skip the two rocprofv3 --pmc passes (roofline.traffic = null)')
    ap.add_argument('--pmc-timeout', type=int, default=240)
    ap.add_argument('--robust-acc-images', type=int, default=256,
                    help='images of the robust-accuracy delta measurement beside the cpu baseline (0: skip; multiples of 64): the '
                         'reference\'s APGD-CE at a fixed L2 bound, HIP vs oracle, paired 95 %% interval (tests/robust_acc_attack.py; '
                         'profiles/ holds a 4096-image run of tools/robust_acc_delta.py apgd)')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip every secondary measurement (rows256, shared encoder, fp32, reference protocol, e4e defender)')
    ap.add_argument('--stub-engine', action='store_true',
                    help='TEST REHEARSAL ONLY: CPU stand-in engines (needs --backend gloo); exercises launcher / collectives / JSON')
    args = ap.parse_args()
    if args.stub_engine and args.backend != 'gloo':
        raise SystemExit('--stub-engine is a CPU rehearsal of the multi-rank plumbing: use it with --backend gloo')

    if args.pmc_child:
        return pmc_child(args)
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N starts them '
                         'itself; under torch.distributed.run pass the same N)')
    pmc_bytes, pmc_note = None, 'skipped'
    args.defender_pmc = {}
    if rank == 0 and world == 1 and not args.no_pmc and not args.stub_engine:
        log('PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over one chunk replay in a child process) ...')
        pmc_bytes, pmc_note = measure_pmc_traffic(args)           # before this process initialises the GPU
        log(f'PMC: {pmc_bytes} bytes per conv launch ({pmc_note[:80]})')
        if not args.no_secondary and args.defender_pmc:
            for w in ('e4e', 'trans'):
                args.defender_pmc[w] = measure_pmc_traffic(args, w)
                log(f'PMC ({w} defender): {args.defender_pmc[w][0]} bytes per conv launch')
    if args.no_secondary:
        args.no_rows256 = args.no_shared_variant = True
    if args.stub_engine:
        device = 'cpu'
    else:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs a GPU: the hot path has no CPU fallback')
        ndev = torch.cuda.device_count()
        dev_index = local_rank if local_rank < ndev else local_rank % ndev      # rehearsal: several ranks share one GPU
        if local_rank >= ndev and args.backend == 'nccl':
            raise SystemExit(f'rank {rank}: no GPU {local_rank} on this node (use --backend gloo only to rehearse)')
        torch.cuda.set_device(dev_index)
        device = f'cuda:{dev_index}'
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
    coll_dev = device if args.backend == 'nccl' else 'cpu'
    if world > 1:
        # torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks: the CPU side of the engine build (weight folding: 12 s on 8
        # threads, 46 s on one) would crawl.  Give each rank of this node its share of the cores (at most 16, the one-GPU share).
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
        torch.set_num_threads(max(1, min(16, cores // max(1, local_world))))

    if args.chunk_rows % args.eot or (args.images * args.eot) % args.chunk_rows:
        raise SystemExit('--chunk-rows must be a multiple of --eot and divide images x eot')
    n_chunks = args.images * args.eot // args.chunk_rows
    n_eng = max(1, min(args.streams, n_chunks))
    rows_per_step = args.images * args.eot
    log(f'rank {rank}/{world}: building weights + {n_eng} engine(s) of {args.chunk_rows} rows')
    if args.stub_engine:
        eng, model = StubEngine(args.chunk_rows, args.eot), None
        engines, streams = [eng], [None]
    else:
        eng, model = build_model(device, args.chunk_rows, args.eot, seed=0, precision=args.precision, share_encoder=args.share_encoder)
        engines = [eng]
        for _ in range(n_eng - 1):
            engines.append(clone_engine(eng, model, device, args))
        streams = [torch.cuda.Stream(device=device) for _ in engines]
    log(f'engines ready: {sum(e.bytes for e in engines) / 1e9:.1f} GB activations + weights, {len(eng.fwd)} fwd + {len(eng.bwd)} bwd ops per chunk, '
        f'{n_chunks} chunks per step')
    g = torch.Generator(device=device).manual_seed(1234 + rank)      # every rank its own images (weak scaling)
    x = torch.rand(args.images, 3, 64, 64, device=device, generator=g)
    # labels = clean prediction of the defender, so that the attack starts from "correct" 
    step = AttackStep(engines, streams, None, x)
    labels = step.forward_only().argmax(dim=1)
    step.labels = labels

    for _ in range(args.warmup):
        step()
    log('warmup done')

    def sync():
        if not args.stub_engine:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if not args.stub_engine:
                torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logits = step()
    # the path's only collective: accuracy counters of every rank (reference: test_defense.py:240-248)
    correct = (logits.argmax(dim=1) == labels).sum().view(1).float()
    counters = torch.stack([correct.squeeze(0), torch.tensor(float(labels.numel()), device=device)])
    if dist is not None:
        counters = counters.to(coll_dev)
        gathered = [torch.zeros_like(counters) for _ in range(world)]
        dist.all_gather(gathered, counters)
        counters = torch.stack(gathered).sum(dim=0)
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=coll_dev if dist is not None else device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0 and int(round(float(counters[1].item()))) != args.images * world:
        # every rank contributes its image count to the gathered counters: anything else means a rank's result went missing
        raise SystemExit(f'accuracy counters cover {float(counters[1].item()):.0f} images, expected {args.images} x {world} ranks')
    if rank == 0 and args.stub_engine:
        print(json.dumps({'metric': 'purified images/sec (attack+encode+decode)', 'value': rows_per_step * world * args.steps / dt,
                          'unit': 'defender rows/s (rows = images x EoT-32)', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                          'dtype': 'none', 'data': 'stub (CPU rehearsal of the multi-rank plumbing; nothing of the path computed)',
                          'config': {'workload': 'stub', 'parallelism': f'image-sharded x{world}'},
                          'accuracy_counters': [float(counters[0].item()), float(counters[1].item())]}))
    elif rank == 0:
        log(f'timed region: {dt:.3f} s for {args.steps} steps')
        rows_total = rows_per_step * world * args.steps
        # ---- roofline of the dominant kernel (conv_mfma_kernel): HIP events on the plan's stream, per launch
        s = eng.stream()
        f_ms, fc_ms, fn = eng.fwd.time(s, iters=1, per_conv=True)
        b_ms, bc_ms, bn = eng.bwd.time(s, iters=1, per_conv=True)
        flops = conv_algorithmic_flops(eng.fwd) + conv_algorithmic_flops(eng.bwd)
        conv_s = (fc_ms + bc_ms) / 1e3
        achieved = flops / conv_s / 1e12
        peak = PEAK_BF16X3_TFLOPS if args.precision == 'bf16x3' else PEAK_FP32_MFMA_TFLOPS
        try:
            mp = measured_peaks(device)
        except Exception as ex:
            mp = {'hbm_copy_tbps': None, 'bf16_mfma_tflops': None, 'how': f'failed: {ex}'}
        meas_ceiling = mp['bf16_mfma_tflops'] / 3.0 if (mp['bf16_mfma_tflops'] and args.precision == 'bf16x3') else None
        out = {
            'metric': 'purified images/sec (attack+encode+decode)',
            'value': rows_total / dt,
            'unit': 'defender rows/s (rows = images x EoT-32)',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 storage; contractions as 3x bf16 MFMA (hi/lo split), fp32 accumulate' if args.precision == 'bf16x3' else 'f32',
            'data': 'synthetic',
            'config': {'workload': 'configs[1]: NVAE purify + VGG-11, 64x64, PGD-Linf step (fwd + input-grad), '
                                   f'bs {args.images} images/GPU x EoT {args.eot} = {rows_per_step} defender rows per step, '
                                   'alphas ours_cosine_no_preprocessing_ids.yaml x0.7, assumed NVAE config (C=32, 3x8 groups, 20 latents)',
                       'images_per_gpu': args.images, 'eot': args.eot, 'rows_per_gpu': rows_per_step,
                       'chunk_rows': args.chunk_rows, 'streams': n_eng,
                       'encoder_shared_by_eot_replicas': bool(eng.share_encoder), 'images_per_step': args.images * world,
                       'images_per_s': args.images * world * args.steps / dt,
                       'parallelism': f'image-sharded x{world}'},
            'roofline': {'bound': 'mfma',
                         'kernel': 'ga::conv_bf3_kernel + ga::conv_halo3_kernel + ga::conv_thin3_kernel + ga::conv_mfma_kernel (implicit-GEMM conv, all instantiations)',
                         'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak, 'traffic': pmc_bytes,
                         'measured_peak': dict(mp, bf16x3_ceiling_tflops=meas_ceiling),
                         'frac_of_measured_peak': (achieved / meas_ceiling) if meas_ceiling else None,
                         'traffic_note': 'HBM bytes per conv launch: ' + pmc_note + '; algorithmic bytes per launch beside it',
                         'algorithmic_bytes_per_launch': (conv_algorithmic_bytes(eng.fwd) + conv_algorithmic_bytes(eng.bwd)) / (fn + bn),
                         'peak_note': ('dense bf16 MFMA peak 2500 TFLOP/s / 3 MFMAs per fp32-class product; achieved counts '
                                       'algorithmic (1x) flops' if args.precision == 'bf16x3' else 'fp32 MFMA peak'),
                         'achieved_over_fp32_mfma_peak': achieved / PEAK_FP32_MFMA_TFLOPS,
                         'launches_per_chunk': int(fn + bn), 'avg_launch_ms': (fc_ms + bc_ms) / (fn + bn),
                         'algorithmic_gflop_per_chunk': flops / 1e9, 'chunks_per_step': n_chunks,
                         'conv_ms_per_chunk': fc_ms + bc_ms, 'plan_ms_fwd': f_ms, 'plan_ms_bwd': b_ms,
                         'note': 'per-launch durations: HIP events around every conv launch of one chunk (forward + backward '
                                 'plan) on one stream, right after the timed region'},
            'accuracy_counters': [float(counters[0].item()), float(counters[1].item())],
        }
        # the second kernel family of the path: the fused decoder cells (ga_dec_cell), per-op HIP events of one chunk
        from gen_adversarial_amd import _lib as L
        cell_ms, cell_n = 0.0, 0
        headline_fused = sum(isinstance(dsc, (L.DecCellDesc, L.DecCellHaloDesc)) for plan in (eng.fwd, eng.bwd) for dsc in plan.descs)
        for plan in (eng.fwd, eng.bwd):
            for dsc, ms in zip(plan.descs, plan.profile(s)):
                if isinstance(dsc, (L.DecCellDesc, L.DecCellHaloDesc)):
                    cell_ms, cell_n = cell_ms + ms, cell_n + 1
        if cell_n:
            cfl = dec_cell_algorithmic_flops(eng.fwd) + dec_cell_algorithmic_flops(eng.bwd)
            out['fused_decoder_cells'] = {
                'kernel': 'ga::dec_cell_fwd_kernel + ga::dec_cell_bwd_kernel (1x1 expand -> SiLU -> depthwise 5x5 -> SiLU -> 1x1 project, '
                          'one launch per direction; the 6C-wide tensors stay in LDS / registers)',
                'launches_per_chunk': cell_n, 'ms_per_chunk': cell_ms, 'avg_launch_ms': cell_ms / cell_n,
                'algorithmic_gflop_per_chunk': cfl / 1e9, 'achieved_tflops': cfl / (cell_ms / 1e3) / 1e12,
                'bound': 'vector ALU + LDS of one wave per SIMD (two quarter-rate transcendentals per SiLU, 25 FMAs and 2.7 LDS reads per '
                         'depthwise output); the contractions are 25 % of its clocks (tools/dec_cell_trace.py)',
                'share_of_plan_ms': cell_ms / (f_ms + b_ms)}
        try:
            out['hbm_bound_kernel_classes'] = dict(
                hbm_bound_classes(eng, s), _note='BASELINE.md §3: achieved_hbm = algorithmic bytes (inputs + outputs of the op, fp32) / '
                f'HIP-event kernel time / 8 TB/s, per memory-bound kernel class of one {args.chunk_rows}-row chunk (forward + backward plan, one stream)')
        except Exception as ex:
            out['hbm_bound_kernel_classes'] = {'_note': f'failed: {type(ex).__name__}: {ex}'}
        if world == 1 and not args.no_secondary and not args.stub_engine:
            # BASELINE.json configs[3] names "PGD-40 + BPDA": the same 8192-row step with the purifier's Jacobian replaced by the
            # identity in the backward pass (forward through purifier + classifier, backward through the classifier alone)
            try:
                st_b = AttackStep(engines, streams, labels, x, bpda=True)
                tb = _time_steps(st_b, 3, warm=1)
                out.setdefault('secondary', {})['pgd_bpda'] = {
                    'rows_per_s': rows_per_step / tb, 'ms_per_step': tb * 1e3,
                    'what': f'--attack pgd-bpda at the headline configuration: {args.images} images x EoT {args.eot} per step, full forward, '
                            'backward through the VGG classifier only (Engine.backward(identity_purifier=True))'}
                del st_b
            except Exception as ex:
                out.setdefault('secondary', {})['pgd_bpda'] = {'rows_per_s': None, 'what': f'failed: {ex}'}
        if world == 1 and rows_per_step != 256 and not args.no_rows256:
            # SURVEY.md §8(d) words configs[1] as R = 256 defender rows (8 images x EoT 32) in ONE plan run: the same
            # attack step at that size, one engine, one stream, reported beside the headline (never as `value`)
            try:
                n8 = 256 // args.eot
                eng8, _ = build_model(device, 256, args.eot, seed=0, precision=args.precision,
                                      share_encoder=args.share_encoder, store=eng.store)
                step8 = AttackStep([eng8], [torch.cuda.Stream(device=device)], labels[:n8].clone(), x[:n8].clone())
                for _ in range(3):
                    step8()
                torch.cuda.synchronize()
                t8 = time.perf_counter()
                for _ in range(20):
                    step8()
                torch.cuda.synchronize()
                t8 = (time.perf_counter() - t8) / 20
                out['config']['rows256_single_plan'] = {'rows_per_s': 256 / t8, 'ms_per_step': t8 * 1e3,
                                                        'what': f'{n8} images x EoT {args.eot} = 256 rows per step, 1 stream'}
                del step8, eng8
            except Exception as ex:
                out['config']['rows256_single_plan'] = {'rows_per_s': None, 'what': f'failed: {ex}'}
        store = eng.store
        if world == 1 and not args.no_shared_variant and not eng.share_encoder and eng.noise_eps == 0.0 and args.eot > 1:
            # the defender API's default (NVAEDefenseModel): with no input noise configured the encoder pass is the same for
            # the EoT replicas of an image and runs once per image — identical numbers, fewer FLOPs.  Reported beside the
            # headline (which runs the literal x.repeat(eot) path), never as `value`.
            try:
                n_chunk_s = min(n_chunks, 4)
                step = None
                engines.clear()
                del eng
                free_gpu_memory()
                es = build_model(device, args.chunk_rows, args.eot, seed=0, precision=args.precision, share_encoder=True, store=store)[0]
                eng_s = [es] + [Engine_clone(es, model, device, args) for _ in range(n_eng - 1)]
                imgs = n_chunk_s * args.chunk_rows // args.eot
                st = AttackStep(eng_s, streams, labels[:imgs].clone(), x[:imgs].clone())
                st()
                torch.cuda.synchronize()
                ts = time.perf_counter()
                for _ in range(3):
                    st()
                torch.cuda.synchronize()
                ts = (time.perf_counter() - ts) / 3
                out['config']['shared_encoder_variant'] = {
                    'rows_per_s': n_chunk_s * args.chunk_rows / ts,
                    'what': f'{imgs} images x EoT {args.eot} per step, encoder once per image (exact without input noise); API default'}
                del st, eng_s, es
                free_gpu_memory()
            except Exception as ex:
                out['config']['shared_encoder_variant'] = {'rows_per_s': None, 'what': f'failed: {ex}'}
        if world == 1 and not args.no_secondary:
            # the headline engines are no longer needed: free their 134 GB before the other secondary measurements
            step = None
            engines.clear()
            eng = None
            free_gpu_memory()
            secondary_measurements(out, args, device, model, store, x, labels)
        if not args.no_cpu_baseline and world == 1:
            try:
                log('cpu baseline (oracle on host cores) ...')
                def parity(xc, epsc, lc, gc):
                    # the same 128 rows on the HIP path (oracle as the checker): logits and input gradient of the CE loss.  The
                    # headline's chunk plans run 32 decoder cells per direction as fused ga_dec_cell launches; at 128 rows the
                    # engine would pick the three unfused launches (fewer than 160 workgroups), so the gate is forced here: the
                    # CHECKED path is the TIMED path (`fused_cells` says how many of the replay's launches were ga_dec_cell)
                    from gen_adversarial_amd.engine import Engine
                    from gen_adversarial_amd import _lib as L_
                    gate = Engine.fuse_min_workgroups
                    Engine.fuse_min_workgroups = 0
                    try:
                        e = build_model(device, xc.shape[0] * args.eot, args.eot, seed=0, precision=args.precision, store=store)[0]
                    finally:
                        Engine.fuse_min_workgroups = gate
                    n_fused = sum(isinstance(d_, (L_.DecCellDesc, L_.DecCellHaloDesc)) for pl in (e.fwd, e.bwd) for d_ in pl.descs)
                    e.x_in.copy_(xc.to(device))
                    for b_, e_ in zip(e.eps, epsc):
                        b_.copy_(e_.to(device))
                    e.forward()
                    lg = e.logits.view(-1, args.eot, e.logits.shape[-1]).mean(dim=1)
                    p = torch.softmax(lg, dim=1)
                    p[torch.arange(p.shape[0], device=p.device), lg.argmax(dim=1)] -= 1.0
                    e.dlogits.view(-1, args.eot, p.shape[-1]).copy_((p / args.eot).unsqueeze(1).expand(-1, args.eot, -1))
                    e.backward()
                    ref_mean = lc.view(-1, args.eot, lc.shape[-1]).mean(dim=1)
                    gd = (e.dx.cpu() - gc).double()
                    out['parity_vs_oracle'] = {
                        'rows': int(lc.shape[0]), 'max_abs_logit_err': float((e.logits.cpu() - lc).abs().max()),
                        'argmax_agree': int((lg.argmax(dim=1).cpu() == ref_mean.argmax(dim=1)).sum()), 'images': int(ref_mean.shape[0]),
                        'input_grad_rel_l2': float(gd.norm() / gc.double().norm()),
                        'fused_cells': int(n_fused),
                        'headline_fused_cells': int(headline_fused),
                        'note': 'same inputs and latent noise as the cpu_baseline sample, replayed with the fused decoder cells forced '
                                'on (the kernels the timed chunk plans select); tolerance of the path: 1e-3 on logits'}
                out['cpu_baseline'] = cpu_baseline(model, 4 * args.eot, args.eot, check=parity)      # 128 rows: ~16 s of oracle time on 16 threads
                log('cpu baseline done')
                if args.robust_acc_images > 0:
                    # robust accuracy under the reference's APGD-CE, HIP vs oracle, every draw pinned, with the paired 95 % interval of
                    # the difference (the oracle as the checker, on a reduced model it can run hundreds of attacks on:
                    # tests/robust_acc_attack.py; VERDICT r03 "next round" #3)
                    log(f'robust accuracy under APGD-CE on {args.robust_acc_images} images (oracle on the host cores) ...')
                    try:
                        sys.path.insert(0, os.path.join(ROOT, 'tests'))
                        from robust_acc_attack import robust_accuracy_under_attack
                        free_gpu_memory()
                        out['robust_accuracy_delta'] = robust_accuracy_under_attack(device, n_images=args.robust_acc_images, eot=2, n_iter=5,
                                                                                    bound=2.0, chunk_images=min(64, args.robust_acc_images))
                    except Exception as ex:
                        out['robust_accuracy_delta'] = {'delta': None, 'what': f'failed: {type(ex).__name__}: {ex}'}
                    log('robust-accuracy delta done')
            except Exception as ex:   # the baseline is a reported number, never a reason to lose the bench line
                out['cpu_baseline'] = {'value': None, 'unit': 'rows/s', 'cores': 0, 'kind': 'port', 'sample': f'failed: {ex}'}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
