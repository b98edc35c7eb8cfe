"""Ray-kernel time of the bench scene as a function of the frameCount uniform (it rotates the bounce sample
hammersley(depth*64 + frameCount, 64) shared by all pixels, raytracingCs.glsl:557): fixed values, then a
free-running counter (what the reference does while TAA is enabled)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opengl_raytracing_amd import host, scenes
rt = host.RayTracer(0)
sc = scenes.make_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 2, host.generate_aabb)
rt.load(sc)
p = sc.params(width=1920, height=1080)
for fc in (0, 1, 2, 3, 5, 7, 16, 32, 48):
    p.frameCount = fc
    ts = []
    for k in range(40):
        rt.render(p); rt.sync(); ts.append(rt.last_kernel_ms())
    print(f"frameCount {fc:3d}: kernel {np.median(ts[10:]):.4f} ms  rays {rt.count_rays(p)}", flush=True)
ts = []
for k in range(200):
    p.frameCount = k
    rt.render(p); rt.sync(); ts.append(rt.last_kernel_ms())
print(f"free-running 0..199: median {np.median(ts[20:]):.4f} ms  min {min(ts[20:]):.4f}  max {max(ts[20:]):.4f}")
