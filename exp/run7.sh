mkdir -p gpurun_out/r3
L=opengl_raytracing_amd/librt_mi355.so
timeout -k 10 200 python tools/gpu_predictor.py 2,4 > gpurun_out/r3/pred3.log 2>&1; grep -E "^C|list-sched" gpurun_out/r3/pred3.log
timeout -k 10 300 python tools/gpu_try.py "pred:@$L" --cfgs=2,3,4,5 --reps=5 > gpurun_out/r3/ab7.log 2>&1; cat gpurun_out/r3/ab7.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_frame.py -x -q -m gpu > gpurun_out/r3/t_parity7.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r3/t_parity7.log
cat > /tmp/loop.py <<'PY'
import sys; sys.path.insert(0, '.')
from opengl_raytracing_amd import host, scenes, layout as L
sc = scenes.make_scene(2, host.generate_aabb); p = sc.params(); rt = host.RayTracer(0); rt.load(sc)
for k in range(60): rt.render(L.copy_params(p, frameCount=k))
rt.sync()
PY
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/prof_pred3 -- python3 /tmp/loop.py > gpurun_out/r3/prof_pred3.log 2>&1
