"""GPU-box tool: regenerates the rocprofv3 evidence of a round under gpurun_out/prof_<tag>/ and writes the summaries that get
committed under profiles/ (this process never touches the GPU itself; every profiled program is `python3 bench.py ...` right
after `--`):
  1. `rocprofv3 --kernel-trace --stats` of the bench command, one stream and the default two streams  -> kernel families (md + csv)
  2. `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE` over
     one chunk replay (`bench.py --pmc-child`)                                                         -> MFMA-pipe utilisation per family
  3. `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) over the same replay           -> HBM bytes per launch per family
    python tools/collect_profiles.py <tag>        (e.g. r02)"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'rXX'
OUT = os.path.join(ROOT, 'gpurun_out', f'prof_{tag}')
os.makedirs(OUT, exist_ok=True)
ROCPROF = '/opt/rocm/bin/rocprofv3'
BENCH = [sys.executable.replace('python', 'python3') if False else 'python3', os.path.join(ROOT, 'bench.py')]
env = dict(os.environ, TMPDIR='/tmp')


def family(n):
    return ('ga::conv_bf3_kernel' if 'conv_bf3_kernel' in n else 'ga::conv_halo3_kernel' if 'conv_halo3' in n else 'ga::conv_thin3_kernel' if 'conv_thin3' in n else
            'ga::conv_mfma_kernel' if 'conv_mfma_kernel' in n else 'ga::conv_splitk_reduce_kernel' if 'splitk' in n else
            'ga::dec_cell_* (fused decoder cell)' if 'dec_cell' in n else 'ga::dwconv5_kernel' if 'dwconv5' in n else 'ga::se_* (excite / apply)' if 'ga::se_' in n else
            n.split('(')[0].replace('void ', '')[:60])


def run(cmd, log):
    print('+', ' '.join(cmd), flush=True)
    with open(log, 'w') as f:
        return subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, env=env, cwd='/tmp').returncode


md = [f'# rocprofv3 evidence, {tag} (tools/collect_profiles.py)\n']
# ---- 1. kernel-trace stats
for name, extra in (('1stream', ['--streams', '1']), ('2streams', [])):
    d = os.path.join(OUT, name)
    shutil.rmtree(d, ignore_errors=True)
    bj = os.path.join(OUT, f'bench_{name}.json')
    cmd = [ROCPROF, '--kernel-trace', '--stats', '--output-format', 'csv', '-d', d, '-o', name, '--'] + BENCH + \
          ['--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-secondary', '--no-pmc'] + extra
    print('+', ' '.join(cmd), flush=True)
    with open(bj, 'w') as fo, open(os.path.join(OUT, f'bench_{name}.err'), 'w') as fe:
        rc = subprocess.run(cmd, stdout=fo, stderr=fe, env=env, cwd='/tmp').returncode
    stats = glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)
    if rc != 0 or not stats:
        md.append(f'## {name}: FAILED (rc {rc})\n')
        continue
    shutil.copy(stats[0], os.path.join(OUT, f'{tag}_bench_kernel_stats_{name}.csv'))
    rows = list(csv.DictReader(open(stats[0])))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    grp = collections.OrderedDict()
    for r in rows:
        g = grp.setdefault(family(r['Name']), [0, 0.0])
        g[0] += int(r['Calls'])
        g[1] += float(r['TotalDurationNs'])
    line = [ln for ln in open(bj).read().splitlines() if ln.startswith('{')]
    b = json.loads(line[-1]) if line else {}
    md.append(f'## bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-pmc {" ".join(extra)} under --kernel-trace --stats\n')
    if b:
        md.append(f'bench line under the profiler: {b["value"]:.0f} rows/s, roofline.avg_launch_ms {b["roofline"]["avg_launch_ms"]:.4f}, '
                  f'achieved {b["roofline"]["achieved"]:.1f} TFLOP/s (frac {b["roofline"]["frac"]:.4f})\n')
    md.append('| kernel family | calls | total ms | avg us | % of kernel time |\n|---|---|---|---|---|')
    for k, (c, t) in sorted(grp.items(), key=lambda kv: -kv[1][1])[:16]:
        md.append(f'| `{k}` | {c} | {t / 1e6:.1f} | {t / c / 1e3:.1f} | {100 * t / tot:.1f} |')
    conv = sum(v[1] for k, v in grp.items() if k.startswith('ga::conv_'))
    nconv = sum(v[0] for k, v in grp.items() if k in ('ga::conv_bf3_kernel', 'ga::conv_halo3_kernel', 'ga::conv_thin3_kernel', 'ga::conv_mfma_kernel'))
    md.append(f'\nconv launches (bf3 + halo3 + thin3 + mfma; split-K reduce time included): {nconv} launches, {conv / max(nconv, 1) / 1e3:.1f} us average\n')
    shutil.rmtree(d, ignore_errors=True)                       # the raw trace is tens of MB: keep the stats only

# ---- 2 + 3. PMC passes over one chunk replay
CHUNK_ROWS = 1024                                 # bench.py's default chunk
child = BENCH + ['--pmc-child', '--chunk-rows', str(CHUNK_ROWS), '--eot', '32']
summary = {}
for passname, counters in (('mfma', ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'GRBM_GUI_ACTIVE']),
                           ('fetch', ['FETCH_SIZE']), ('write', ['WRITE_SIZE'])):
    d = os.path.join(OUT, 'pmc_' + passname)
    shutil.rmtree(d, ignore_errors=True)
    rc = run([ROCPROF, '--pmc'] + counters + ['--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'pmc', '--'] + child,
             os.path.join(OUT, f'pmc_{passname}.log'))
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for f in files:
        for row in csv.DictReader(open(f, newline='')):
            fam = family(row['Kernel_Name'])
            fam = fam if fam.startswith('ga::') else 'other'
            acc[fam][row['Counter_Name']] += float(row['Counter_Value'])
            launches[fam].add(row['Dispatch_Id'])
    summary[passname] = {fam: dict(v, launches=len(launches[fam])) for fam, v in acc.items()}
    summary[passname]['_rc'] = rc
    shutil.rmtree(d, ignore_errors=True)
res = {'_how': 'rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --pmc-child --chunk-rows 1024 --eot 32 (one forward + backward replay of a '
               '512-row chunk plan, weight preparation kernels included under "other"); counters summed over the launches of each kernel family. '
               'mfma_pipe_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs).  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 '
               'FETCH_SIZE counts half of the bytes of wide coalesced reads: hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches.'}
for fam, v in summary.get('mfma', {}).items():
    if fam.startswith('_'):
        continue
    g = v.get('GRBM_GUI_ACTIVE', 0.0)
    v['mfma_pipe_utilisation'] = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (g / 8.0 * 1024.0) if g else None
    w = v.get('SQ_WAVE_CYCLES', 0.0)
    if w:
        v['frac_wait_any'] = v.get('SQ_WAIT_ANY', 0.0) / w
        v['frac_wait_inst'] = v.get('SQ_WAIT_INST_ANY', 0.0) / w
res['mfma'] = summary.get('mfma')
traffic = {}
for fam in set(summary.get('fetch', {})) | set(summary.get('write', {})):
    if fam.startswith('_'):
        continue
    f_, w_ = summary['fetch'].get(fam, {}), summary['write'].get(fam, {})
    n = max(f_.get('launches', 0), w_.get('launches', 0), 1)
    traffic[fam] = {'launches': n, 'FETCH_SIZE_KB': f_.get('FETCH_SIZE', 0.0), 'WRITE_SIZE_KB': w_.get('WRITE_SIZE', 0.0),
                    'hbm_bytes_per_launch': (2 * f_.get('FETCH_SIZE', 0.0) + w_.get('WRITE_SIZE', 0.0)) * 1024.0 / n}
conv = [v for k, v in traffic.items() if k in ('ga::conv_bf3_kernel', 'ga::conv_halo3_kernel', 'ga::conv_thin3_kernel', 'ga::conv_mfma_kernel')]
if conv:
    traffic['conv_kernels'] = {'launches': sum(v['launches'] for v in conv),
                               'hbm_bytes_per_launch': sum(v['hbm_bytes_per_launch'] * v['launches'] for v in conv) / sum(v['launches'] for v in conv)}
traffic['all_kernels_total_bytes'] = sum(v['hbm_bytes_per_launch'] * v['launches'] for k, v in traffic.items() if k.startswith('ga::') or k == 'other')
traffic['hot_path_kernels_total_bytes'] = sum(v['hbm_bytes_per_launch'] * v['launches'] for k, v in traffic.items() if k.startswith('ga::'))
traffic['rows'] = CHUNK_ROWS
traffic['hbm_bytes_per_attack_row'] = traffic['hot_path_kernels_total_bytes'] / float(CHUNK_ROWS)
traffic['_note'] = ("'other' = PyTorch kernels of engine construction (zero-filling the activation buffers, weight upload / bf16 split): not on the "
                    "hot path; hbm_bytes_per_attack_row counts the ga:: kernels only")
res['traffic'] = traffic

# ---- 4. the configs[2] / configs[4] defenders: one forward + backward replay of the plan bench.py times (32-row e4e plan, 64-row
#         Style-Transformer plan): kernel-trace stats by family, then FETCH_SIZE / WRITE_SIZE passes by family
for w, rows, what in (('e4e', 32, 'configs[2]: IR-SE50 e4e @256 px -> StyleGAN2-1024 -> ResNet-50, 32-row plan, initial_noise_eps 4.0'),
                      ('trans', 64, 'configs[4]: blur -> Style-Transformer encoder -> StyleGAN2-512 -> ResNeXt-50 @128 px, 64-row plan')):
    wchild = BENCH + ['--pmc-child', '--pmc-workload', w, '--eot', '32']
    d = os.path.join(OUT, f'{w}_trace')
    shutil.rmtree(d, ignore_errors=True)
    rc = run([ROCPROF, '--kernel-trace', '--stats', '--output-format', 'csv', '-d', d, '-o', w, '--'] + wchild, os.path.join(OUT, f'{w}_trace.log'))
    stats = glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)
    md.append(f'## {what}: one forward + backward replay (`bench.py --pmc-child --pmc-workload {w}`) under --kernel-trace --stats\n')
    if rc != 0 or not stats:
        md.append(f'FAILED (rc {rc})\n')
        continue
    shutil.copy(stats[0], os.path.join(OUT, f'{tag}_{w}_defender_kernel_stats.csv'))
    rows_ = [r for r in csv.DictReader(open(stats[0])) if 'ga::' in r['Name']]
    tot = sum(float(r['TotalDurationNs']) for r in rows_)
    grp = collections.OrderedDict()
    for r in rows_:
        g = grp.setdefault(family(r['Name']), [0, 0.0])
        g[0] += int(r['Calls'])
        g[1] += float(r['TotalDurationNs'])
    shutil.rmtree(d, ignore_errors=True)
    tr = {}
    for passname, counter in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
        d = os.path.join(OUT, f'{w}_pmc_{passname}')
        shutil.rmtree(d, ignore_errors=True)
        run([ROCPROF, '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'pmc', '--'] + wchild,
            os.path.join(OUT, f'{w}_pmc_{passname}.log'))
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for row in csv.DictReader(open(f, newline='')):
                if 'ga::' in row['Kernel_Name'] and row['Counter_Name'] == counter:
                    tr.setdefault(family(row['Kernel_Name']), {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0})[counter] += float(row['Counter_Value'])
        shutil.rmtree(d, ignore_errors=True)
    md.append(f'ga:: kernels only (engine construction excluded): {tot / 1e6:.1f} ms for {rows} rows\n')
    md.append('| kernel family | calls | total ms | avg us | % of kernel time | HBM MB (2 x FETCH_SIZE + WRITE_SIZE) | TB/s while running |\n|---|---|---|---|---|---|---|')
    allb = 0.0
    for k, (c, t) in sorted(grp.items(), key=lambda kv: -kv[1][1])[:14]:
        b = (2 * tr.get(k, {}).get('FETCH_SIZE', 0.0) + tr.get(k, {}).get('WRITE_SIZE', 0.0)) * 1024.0
        md.append(f'| `{k}` | {c} | {t / 1e6:.1f} | {t / c / 1e3:.1f} | {100 * t / tot:.1f} | {b / 1e6:.0f} | {b / t / 1e3 if t else 0:.2f} |')
    allb = sum((2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024.0 for v in tr.values())
    md.append(f'\nHBM traffic of the replay: {allb / 1e9:.1f} GB = {allb / rows / 1e6:.0f} MB per attack row\n')
    res[f'{w}_defender_traffic'] = {'rows': rows, 'hbm_bytes_per_attack_row': allb / rows,
                                    'by_family': {k: (2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024.0 for k, v in tr.items()}}
json.dump(res, open(os.path.join(OUT, f'{tag}_pmc_summary.json'), 'w'), indent=1)
open(os.path.join(OUT, f'{tag}_kernel_families.md'), 'w').write('\n'.join(md) + '\n')
print('\n'.join(md))
print(json.dumps({k: v for k, v in res['traffic'].items() if not isinstance(v, dict)}, indent=1))
