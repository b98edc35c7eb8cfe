"""Packet-coherence diagnostics of the packet kernel per config (instrumented build, rt_debug_stats_ex).
usage: python tools/gpu_packet_stats.py [cfgs=2,3,4,5] [WxH]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opengl_raytracing_amd import host, scenes
rt = host.RayTracer(0)
size = sys.argv[2] if len(sys.argv) > 2 else None
for cfg in [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "2,3,4,5").split(",")]:
    sc = scenes.make_scene(cfg, host.generate_aabb)
    w, h = (int(v) for v in size.split("x")) if size else (sc.width, sc.height)
    p = sc.params(width=w, height=h)
    rt.load(sc)
    for mode, fn in (("reference rays", rt.count_rays), ("traced rays", rt.count_rays_traced)):
        rays = fn(p)
        s = rt.debug_stats_ex()
        pk = max(s[1], 1)
        print(f"C{cfg} {w}x{h} [{mode}]: rays {rays}  packets {s[1]}  lanes/packet {s[11] / pk:.1f}  candidates/packet {s[2] / pk:.2f} of {len(sc.objects)}  "
              f"cull passes/packet {s[3] / pk:.2f} | usable axes 3/2/1/0: {s[4] / pk:.3f} {s[5] / pk:.3f} {s[6] / pk:.3f} {s[7] / pk:.3f}  uncullable {s[8] / pk:.4f} | "
              f"cand/packet with 3 axes {s[9] / max(s[4], 1):.2f}, with fewer {s[10] / max(s[5] + s[6] + s[7] + s[8], 1):.2f}", flush=True)
