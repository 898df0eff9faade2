python -m pytest tests/test_ops_gpu.py -x -q -k "thin3" > gpurun_out/r04_t6.log 2>&1; tail -5 gpurun_out/r04_t6.log
for sh in "512 64 32 32" "1024 64 32 32"; do
 for lib in libga_ops libga_ops_thin_staged; do
  export GA_OPS_LIB=$PWD/gen_adversarial_amd/$lib.so
  echo "== $lib"
  python tools/conv_ab.py $sh 1 7 11 2>&1 | grep -v amdgpu.ids
  GA_AB_BWD=1 python tools/conv_ab.py $sh 0 7 11 2>&1 | grep -v amdgpu.ids
 done
done > gpurun_out/r04_thin_ab2.log 2>&1
cat gpurun_out/r04_thin_ab2.log
