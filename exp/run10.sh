mkdir -p gpurun_out/r3
L=opengl_raytracing_amd/librt_mi355.so
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3/t_all10.log 2>&1; echo "all gpu tests rc=$?"; tail -6 gpurun_out/r3/t_all10.log
timeout -k 10 300 python tools/gpu_try.py "cur:@$L" --cfgs=2,3,4,5 --reps=5 > gpurun_out/r3/ab10.log 2>&1; cat gpurun_out/r3/ab10.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3/bench10.json 2> gpurun_out/r3/bench10.err; echo "bench rc=$?"; tail -3 gpurun_out/r3/bench10.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench10.json').read().strip().splitlines()[-1])
for k in ['value','ms_per_step','step_ms_device','frame_ms','free_running','cold_frame_ms','raster_order_ms','pipelined','cpu_baseline']:
    print(k, d.get(k))
for c,v in d.get('configs_extra',{}).items(): print(c, v['ms_per_step'], v['frame_ms'])
PY
