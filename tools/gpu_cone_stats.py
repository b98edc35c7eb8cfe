"""Candidates entering / leaving the per-lane cull level of pk_pcf_shadow (instrumented launch, rt_debug_stats_ex [20]/[21])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opengl_raytracing_amd import host, scenes
rt = host.RayTracer(0)
for cfg in [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "4,5").split(",")]:
    sc = scenes.make_scene(cfg, host.generate_aabb)
    for size in ((1920, 1080), (sc.width, sc.height)):
        p = sc.params(width=size[0], height=size[1])
        rt.load(sc)
        rays = rt.count_rays_traced(p); s = rt.debug_stats_ex()
        print(f"C{cfg} {size[0]}x{size[1]}: traced rays {rays}  light-level candidates {s[20]} -> after the lane level {s[21]} ({s[21] / max(s[20], 1):.3f})", flush=True)
