"""Where does a frame's time go?  Same scene with the shadow rays switched off (shadowType 0), with 1 PCF sample, and as is.
usage: python tools/gpu_breakdown.py [cfgs=4,5]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opengl_raytracing_amd import host, scenes
rt = host.RayTracer(0)
for cfg in [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "4,5").split(",")]:
    for mode in ("as is", "pcf samples 1", "no shadows", "no shadows, depth 1"):
        sc = scenes.make_scene(cfg, host.generate_aabb)
        if mode == "pcf samples 1": sc.lights["pcfSamples"] = 1
        if mode.startswith("no shadows"): sc.lights["shadowType"] = 0
        p = sc.params(max_ray_depth=1 if mode.endswith("depth 1") else None)
        rt.load(sc)
        for _ in range(3): rt.render(p)
        rt.sync(); t0 = time.perf_counter()
        k = 5
        for _ in range(k): rt.render(p)
        rt.sync(); ms = (time.perf_counter() - t0) / k * 1e3
        rays = rt.count_rays_traced(p); s = rt.debug_stats_ex()
        print(f"C{cfg} {sc.width}x{sc.height} [{mode}]: {ms:.3f} ms  traced rays {rays}  packets {s[1]}  cand/packet {s[2] / max(s[1], 1):.1f}", flush=True)
