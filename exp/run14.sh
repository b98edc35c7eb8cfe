mkdir -p gpurun_out/r3
L=opengl_raytracing_amd/librt_mi355.so
timeout -k 10 400 python tools/gpu_try.py "nores:@exp/lib_nores.so" "res:@$L" "nores2:@exp/lib_nores.so" "res2:@$L" --cfgs=2,3,4,5 --reps=5 > gpurun_out/r3/ab14.log 2>&1; cut -c1-110 gpurun_out/r3/ab14.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_mgpu.py tests/test_frame.py -x -q -m gpu > gpurun_out/r3/t_14.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3/t_14.log
