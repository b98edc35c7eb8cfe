mkdir -p gpurun_out/r3
L=opengl_raytracing_amd/librt_mi355.so
( for g in "16,32,24" "32,64,24" "32,64,48" "64,128,32" "24,48,12" "48,96,32"; do echo "== GEOM $g"; RT_ST_GEOM=$g timeout -k 10 120 python tools/gpu_try.py "tab:@$L" --cfgs=2,4,5 --reps=5; done
echo "== variants"; timeout -k 10 500 python tools/gpu_try.py "tab:@$L" "noreach:@exp/lib_noreach.so" "hw5:@exp/lib_hw5.so" "hb1:@exp/lib_hb1.so" "lg1:@exp/lib_lg1.so" "w6:@exp/lib_w6.so" "hg1:@exp/lib_hg1.so" --cfgs=2,3,4,5 --reps=5 ) > gpurun_out/r3/ab2.log 2>&1
cat gpurun_out/r3/ab2.log
