python -m pytest tests -m gpu -x -q > gpurun_out/r04_t12.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04_t12.log; tail -4 gpurun_out/r04_t12.log
( time python bench.py > gpurun_out/r04_bench_d.json 2> gpurun_out/r04_bench_d.err ) 2>> gpurun_out/r04_bench_d.err
tail -3 gpurun_out/r04_bench_d.err
python -c "
import json
d=json.loads(open('gpurun_out/r04_bench_d.json').read().strip().splitlines()[-1])
s=d['secondary']
print(d['value'], d['roofline']['frac'], d['roofline']['measured_peak']['bf16_mfma_tflops'], s['configs2_e4e_defender']['rows_per_s'], s['configs4_trans_defender']['rows_per_s'], s['reference_protocol_1_image']['rows_per_s_eager'], s['reference_protocol_1_image']['rows_per_s_api_default_shared_encoder'])
"
