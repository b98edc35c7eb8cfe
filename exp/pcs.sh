mkdir -p gpurun_out/r3/pcs
cat > /tmp/loop.py <<'PY'
import sys; sys.path.insert(0, '.')
from opengl_raytracing_amd import host, scenes
sc = scenes.make_scene(2, host.generate_aabb); p = sc.params(); rt = host.RayTracer(0); rt.load(sc)
for _ in range(300): rt.render(p)
rt.sync()
PY
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 -L 2>/dev/null | grep -i -A6 "pc.sampl" | head -30
echo "== stochastic"
timeout -k 10 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method stochastic --pc-sampling-interval 1048576 --output-format csv -d gpurun_out/r3/pcs/st -- python3 /tmp/loop.py > gpurun_out/r3/pcs/st.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r3/pcs/st.log
echo "== host_trap"
timeout -k 10 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 100 --output-format csv -d gpurun_out/r3/pcs/ht -- python3 /tmp/loop.py > gpurun_out/r3/pcs/ht.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r3/pcs/ht.log
find gpurun_out/r3/pcs -type f | head; du -sh gpurun_out/r3/pcs
