for e in "$@"; do
  export GA_OPS_LIB=$PWD/gen_adversarial_amd/libga_ops_exp$e.so
  echo "=== EXP $e"
  for args in "256 16 128 768 1 1 1 0 0 0" "256 16 128 768 1 1 1 0 0 1" "256 16 128 128 3 1 1 0 0 0" "256 16 128 128 3 1 1 0 0 1"; do
    python tools/conv_trace.py $args 2>&1 | grep -v "amdgpu\|grid span\|workgroups per\|start offset"
  done
done
