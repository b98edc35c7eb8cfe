mkdir -p gpurun_out/r3
L=opengl_raytracing_amd/librt_mi355.so
timeout -k 10 300 python tools/gpu_try.py "pred:@$L" --cfgs=2,3,4,5 --reps=5 > gpurun_out/r3/ab3.log 2>&1; cat gpurun_out/r3/ab3.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_frame.py tests/test_taa.py -x -q -m gpu > gpurun_out/r3/t_parity3.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r3/t_parity3.log
