python tools/collect_profiles.py r04 > gpurun_out/r04_collect_b.log 2>&1; echo "collect rc $?"
tail -5 gpurun_out/r04_collect_b.log
python tools/plan_profile.py 1024 > gpurun_out/r04_plan1024_b.log 2>&1 && python tools/plan_profile.py 32 > gpurun_out/r04_plan32_b.log 2>&1
echo "plan rc $?"
