mkdir -p gpurun_out/r3
L=opengl_raytracing_amd/librt_mi355.so
timeout -k 10 400 python tools/gpu_try.py "base94:@exp/lib_base94.so" "cur:@$L" "base94b:@exp/lib_base94.so" "curb:@$L" --cfgs=2,4,5 --reps=5 > gpurun_out/r3/ab11.log 2>&1; cat gpurun_out/r3/ab11.log
