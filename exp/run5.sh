mkdir -p gpurun_out/r3
L=opengl_raytracing_amd/librt_mi355.so
timeout -k 10 200 python tools/gpu_predictor.py 2,4 > gpurun_out/r3/pred2.log 2>&1; grep -E "^C" gpurun_out/r3/pred2.log
timeout -k 10 300 python tools/gpu_try.py "pred:@$L" --cfgs=2,3,4,5 --reps=5 > gpurun_out/r3/ab5.log 2>&1; cat gpurun_out/r3/ab5.log
cat > /tmp/loop.py <<'PY'
import sys; sys.path.insert(0, '.')
from opengl_raytracing_amd import host, scenes
sc = scenes.make_scene(2, host.generate_aabb); p = sc.params(); rt = host.RayTracer(0); rt.load(sc)
for _ in range(60): rt.render(p)
rt.sync()
PY
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/prof_pred2 -- python3 /tmp/loop.py > gpurun_out/r3/prof_pred2.log 2>&1
find gpurun_out/r3/prof_pred2 -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-60,200-300 | head -6
