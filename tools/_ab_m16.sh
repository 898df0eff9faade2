export GA_OPS_LIB=$PWD/gen_adversarial_amd/libga_ops_hexp.so
for sh in "512 16 128 128" "512 8 256 256" "512 4 512 512" "512 32 64 64"; do
  python tools/conv_ab.py $sh 1 8 9 10 2>&1 | grep -v amdgpu.ids
done
