python -m pytest tests -m gpu -x -q > gpurun_out/r04_t3.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04_t3.log
for c in 512 1024 512 1024; do
python bench.py --steps 4 --warmup 2 --no-pmc --no-secondary --no-cpu-baseline --chunk-rows $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk', $c, d['value'], d['roofline']['achieved'])"
done > gpurun_out/r04_chunk_ab.log 2>&1
tail -4 gpurun_out/r04_t3.log; cat gpurun_out/r04_chunk_ab.log
