mkdir -p gpurun_out/r3
L=opengl_raytracing_amd/librt_mi355.so
timeout -k 10 800 python tools/gpu_try.py "cur:@$L" "w6:@exp/lib_w6.so" "hw7:@exp/lib_hw7.so" "g4:@exp/lib_g4.so" "lreach:@exp/lib_lreach.so" "sss2:@exp/lib_sss2.so" "hw5:@exp/lib_hw5.so" "cur_b:@$L" --cfgs=2,4,5 --reps=5 > gpurun_out/r3/ab15.log 2>&1; cut -c1-72 gpurun_out/r3/ab15.log
