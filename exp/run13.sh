mkdir -p gpurun_out/r3
L=opengl_raytracing_amd/librt_mi355.so
( timeout -k 10 400 python tools/gpu_try.py "nogather:@exp/lib_nogather.so" "gather:@$L" "nogather2:@exp/lib_nogather.so" "gather2:@$L" --cfgs=2,3,4,5 --reps=5
for g in "32,64,16" "24,48,16" "48,96,32"; do echo "== GEOM $g"; RT_ST_GEOM=$g timeout -k 10 120 python tools/gpu_try.py "gather:@$L" --cfgs=2 --reps=5; done ) > gpurun_out/r3/ab13.log 2>&1; cut -c1-150 gpurun_out/r3/ab13.log
