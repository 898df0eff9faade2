python -m pytest tests/test_ops_gpu.py -x -q -k "thin3 or halo3_fragment or conv" > gpurun_out/r04_t5.log 2>&1; tail -5 gpurun_out/r04_t5.log
for sh in "512 64 32 32" "128 128 32 32" "1024 64 32 32"; do
  python tools/conv_ab.py $sh 1 7 11 2>&1 | grep -v amdgpu.ids
  GA_AB_BWD=1 python tools/conv_ab.py $sh 0 7 11 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04_thin_ab.log 2>&1
cat gpurun_out/r04_thin_ab.log
