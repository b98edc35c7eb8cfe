mkdir -p gpurun_out/r3
RT_PROFILE_CFG=c2 bash profiles/run_profile.sh r03_packet > gpurun_out/r3/prof_c2.log 2>&1; tail -3 gpurun_out/r3/prof_c2.log
RT_PROFILE_CFG=c5 bash profiles/run_profile.sh r03_c5 --config 5 --steps 10 --warmup 3 --no-cpu-baseline --extra-configs none --no-modes > gpurun_out/r3/prof_c5.log 2>&1; tail -3 gpurun_out/r3/prof_c5.log
mkdir -p gpurun_out/r3/profiles && cp profiles/r03_* profiles/kernel_counters.json gpurun_out/r3/profiles/
