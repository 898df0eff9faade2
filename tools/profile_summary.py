"""Summarises rocprofv3 --kernel-trace --stats CSVs of bench.py per kernel family (markdown on stdout).
usage: python tools/profile_summary.py <dir with s1_/s2_kernel_stats.csv> <bench json 1-stream> <bench json 2-streams>"""
import collections, csv, json, sys

d, b1, b2 = sys.argv[1:4]
print('# rocprofv3 --kernel-trace --stats of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rows256 [--streams 1]` (round 1, v3)\n')
print('Per kernel family (all template instantiations summed); source CSVs: r01_bench_kernel_stats_v3_{1stream,2streams}.csv.\n')
for tag, bj, nm in (('s1', b1, '1 stream (kernels run alone: durations comparable with bench.py roofline.avg_launch_ms)'),
                    ('s2', b2, '2 streams (default; two chunks share the GPU, so per-kernel durations stretch while throughput rises)')):
    rows = list(csv.DictReader(open(f'{d}/{tag}_kernel_stats.csv')))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    grp = collections.OrderedDict()
    for r in rows:
        n = r['Name']
        key = ('ga::conv_bf3_kernel' if 'conv_bf3_kernel' in n else 'ga::conv_halo3_kernel' if 'conv_halo3' in n else
               'ga::conv_mfma_kernel' if 'conv_mfma_kernel' in n else 'ga::conv_splitk_reduce_kernel' if 'splitk' in n else
               'ga::dec_cell_* (fused decoder cell)' if 'dec_cell' in n else 'ga::dwconv5_kernel' if 'dwconv5' in n else 'ga::se_* (excite / apply)' if 'se_' in n else n.split('(')[0])
        g = grp.setdefault(key, [0, 0.0])
        g[0] += int(r['Calls'])
        g[1] += float(r['TotalDurationNs'])
    b = json.load(open(bj))
    print(f'## {nm}\n')
    print(f'bench line under the profiler: {b["value"]:.0f} rows/s, roofline.avg_launch_ms {b["roofline"]["avg_launch_ms"]:.4f}, '
          f'achieved {b["roofline"]["achieved"]:.1f} TFLOP/s\n')
    print('| kernel family | calls | total ms | avg us | % of kernel time |\n|---|---|---|---|---|')
    for k, (c, t) in sorted(grp.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f'| `{k}` | {c} | {t / 1e6:.1f} | {t / c / 1e3:.1f} | {100 * t / tot:.1f} |')
    conv = sum(v[1] for k, v in grp.items() if 'conv_' in k and 'dwconv' not in k)
    nconv = sum(v[0] for k, v in grp.items() if k in ('ga::conv_bf3_kernel', 'ga::conv_halo3_kernel', 'ga::conv_mfma_kernel'))
    print(f'\nconv launches (bf3 + halo3 + mfma, split-K reduce time included): {nconv} launches, {conv / nconv / 1e3:.1f} us average\n')
