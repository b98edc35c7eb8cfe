mkdir -p gpurun_out/r3
L=opengl_raytracing_amd/librt_mi355.so
( for e in 0 1 2; do echo "== RT_SCHED_EXP=$e"; RT_SCHED_EXP=$e timeout -k 10 200 python tools/gpu_try.py "pred:@$L" --cfgs=2,4 --reps=3; done ) > gpurun_out/r3/ab9.log 2>&1; cat gpurun_out/r3/ab9.log
