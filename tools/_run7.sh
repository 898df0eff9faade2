python tools/ref_protocol.py > gpurun_out/r04_ref_protocol_before.json 2>/dev/null; cat gpurun_out/r04_ref_protocol_before.json
python tools/tune.py 32 gpurun_out/r04_tune_32share.json add share > gpurun_out/r04_tune_32share.log 2>&1; tail -2 gpurun_out/r04_tune_32share.log
cp gpurun_out/r04_tune_32share.json gen_adversarial_amd/conv_tune_gfx950.json
python tools/ref_protocol.py > gpurun_out/r04_ref_protocol_after.json 2>/dev/null; cat gpurun_out/r04_ref_protocol_after.json
python tools/robust_acc_delta.py apgd 4096 2 5 2.0 > gpurun_out/r04_robust_acc_apgd_4096.json 2> gpurun_out/r04_robust_acc_apgd_4096.err; cat gpurun_out/r04_robust_acc_apgd_4096.json
python tools/robust_acc_delta.py fullsize 8 2 3 2.0 > gpurun_out/r04_fullsize_verdicts.json 2> gpurun_out/r04_fullsize_verdicts.err; cat gpurun_out/r04_fullsize_verdicts.json
python -m pytest tests/test_attack_parity_gpu.py -q -s > gpurun_out/r04_attack_parity.log 2>&1; tail -3 gpurun_out/r04_attack_parity.log
