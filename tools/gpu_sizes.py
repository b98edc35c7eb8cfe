"""Ad-hoc: C2 scene at several frame sizes (tail / launch-overhead sensitivity)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opengl_raytracing_amd import scenes, host
rt = host.RayTracer(0)
sc = scenes.make_scene(2, host.generate_aabb)
rt.load(sc)
for (w, h) in [(960, 540), (1920, 1080), (3840, 2160), (7680, 4320)]:
    p = sc.params(width=w, height=h)
    rt.render(p); rt.sync()
    ts = []
    for _ in range(5):
        rt.render(p); ts.append(rt.last_kernel_ms())
    rays = rt.count_rays(p)
    print(f"{w}x{h}: {np.median(ts):.3f} ms  {rays/np.median(ts)/1e3:.0f} Mray/s  rays/px {rays/(w*h):.2f}", flush=True)
