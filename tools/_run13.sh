python -m pytest tests -m gpu -x -q > gpurun_out/r04_t7.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04_t7.log; tail -4 gpurun_out/r04_t7.log
( time python bench.py > gpurun_out/r04_bench_b.json 2> gpurun_out/r04_bench_b.err ) 2>> gpurun_out/r04_bench_b.err
tail -3 gpurun_out/r04_bench_b.err
python -c "
import json
d=json.loads(open('gpurun_out/r04_bench_b.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['achieved'], d['secondary']['configs2_e4e_defender']['rows_per_s'], d['secondary']['reference_protocol_1_image'])
"
