python -m pytest tests/test_fullsize_configs_gpu.py tests/test_fullsize_gpu.py tests/test_ops_gpu.py tests/test_resnet_gpu.py tests/test_stylegan_gpu.py tests/test_trans_gpu.py -m gpu -x -q -s > gpurun_out/r04_t4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04_t4.log
tail -3 gpurun_out/r04_t4.log
( time python bench.py > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err ) 2>> gpurun_out/r04_bench_a.err
tail -3 gpurun_out/r04_bench_a.err
python -c "
import json
d=json.loads(open('gpurun_out/r04_bench_a.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d.get('robust_accuracy_delta',{}).get('delta'), d.get('robust_accuracy_delta',{}).get('ci95'))
"
