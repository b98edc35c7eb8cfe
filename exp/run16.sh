mkdir -p gpurun_out/r3
L=opengl_raytracing_amd/librt_mi355.so
timeout -k 10 800 python tools/gpu_try.py "cur:@$L" "sss4:@exp/lib_sss4.so" "hb0:@exp/lib_hb0.so" "hsss2:@exp/lib_hsss2.so" "hhb1:@exp/lib_hhb1.so" "lpcss_res:@exp/lib_lpcss_res.so" "cur_b:@$L" --cfgs=2,3,4,5 --reps=5 > gpurun_out/r3/ab16.log 2>&1; cut -c1-72 gpurun_out/r3/ab16.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3/t_all16.log 2>&1; echo "all gpu tests rc=$?"; tail -3 gpurun_out/r3/t_all16.log
