"""Ad-hoc first-light script for the GPU box (not a pytest file): HIP vs oracle on small frames."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opengl_raytracing_amd import scenes, host
from oracle import binding as O

def cmp(a, b, name):
    bn = np.isnan(a) & np.isnan(b)
    ex = ((a == b) | bn).all(axis=-1)
    e = np.abs(a.astype(np.float64) - b.astype(np.float64)); tol = 1e-4 * np.maximum(np.abs(a), np.abs(b)) + 1e-6
    ok = ((e <= tol) | bn).all(axis=-1)
    print(f"   {name}: exact {ex.mean()*100:.4f}%  pass(1e-4) {ok.mean()*100:.4f}%  fail {int((~ok).sum())}", flush=True)

rt = host.RayTracer(0)
for cfg, (w, h) in [(1, (256, 256)), (2, (480, 270)), (3, (480, 270)), (4, (320, 180)), (5, (320, 180))]:
    sc = scenes.make_scene(cfg, host.generate_aabb)
    p = sc.params(width=w, height=h)
    rt.load(sc)
    rt.render(p); rt.sync()
    t = time.time(); rt.render(p); rt.sync(); ms = rt.last_kernel_ms()
    col, pos, nrm = rt.readback()
    rays = rt.count_rays(p)
    t = time.time(); oc, op, on, orays = O.render(sc, p); cpu = time.time() - t
    print(f"cfg {cfg} {w}x{h}: kernel {ms:.3f} ms, rays {rays} (oracle {orays}), {rays/ms/1e3:.1f} Mray/s; oracle {cpu:.2f}s", flush=True)
    cmp(col, oc, "color"); cmp(pos, op, "pos"); cmp(nrm.astype(np.float32), on.astype(np.float32), "normal")
# full-size C2 timing
sc = scenes.make_scene(2, host.generate_aabb)
p = sc.params()
rt.load(sc)
for i in range(3):
    rt.render(p); rt.sync(); print("C2 full 1080p kernel ms", rt.last_kernel_ms(), flush=True)
rays = rt.count_rays(p)
print("C2 rays", rays, "Mray/s", rays / rt.last_kernel_ms() / 1e3)
