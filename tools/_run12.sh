python tools/retune_thin.py gpurun_out/r04_tune_thin.json > gpurun_out/r04_tune_thin.log 2>&1; tail -40 gpurun_out/r04_tune_thin.log | cut -c1-200
