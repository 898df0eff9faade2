python tools/robust_acc_delta.py apgd 6144 2 5 2.0 1 > gpurun_out/r04_robust_acc_apgd_6144_seed1.json 2> gpurun_out/r04_robust_acc_apgd_seed1.err; cat gpurun_out/r04_robust_acc_apgd_6144_seed1.json | cut -c1-600
python tools/robust_acc_delta.py fullsize 64 2 3 2.0 > gpurun_out/r04_fullsize_verdicts_64.json 2> gpurun_out/r04_fullsize_verdicts_64.err; cat gpurun_out/r04_fullsize_verdicts_64.json
python -m pytest tests/test_attack_parity_gpu.py -q -s > gpurun_out/r04_attack_parity.log 2>&1; tail -3 gpurun_out/r04_attack_parity.log
