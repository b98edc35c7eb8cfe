mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3/t_all12.log 2>&1; echo "all gpu tests rc=$?"; tail -4 gpurun_out/r3/t_all12.log
RT_TIMERS_LIB=exp/lib_timers.so timeout -k 10 200 python tools/gpu_timers.py 2,3,4,5 > gpurun_out/r3/timers12.log 2>&1; cat gpurun_out/r3/timers12.log
bash exp/pcs.sh > gpurun_out/r3/pcs12.log 2>&1; tail -30 gpurun_out/r3/pcs12.log
