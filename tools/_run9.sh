python tools/robust_acc_delta.py apgd 6144 2 5 2.0 1 > gpurun_out/r04_robust_acc_apgd_6144_data1.json 2> gpurun_out/r04_robust_acc_apgd_data1.err; cat gpurun_out/r04_robust_acc_apgd_6144_data1.json | cut -c1-500
python -m pytest tests/test_attack_parity_gpu.py -q -s -k "cw or fab" > gpurun_out/r04_attack_parity_cw.log 2>&1; tail -3 gpurun_out/r04_attack_parity_cw.log
python tools/collect_profiles.py r04 > gpurun_out/r04_collect.log 2>&1; tail -5 gpurun_out/r04_collect.log
