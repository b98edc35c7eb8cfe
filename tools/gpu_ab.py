"""Ad-hoc A/B timing of kernel variants on the GPU box (not a pytest file)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opengl_raytracing_amd import scenes, host

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
cfgs = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "2,3,4,5").split(",")]
rt = host.RayTracer(0)
for cfg in cfgs:
    sc = scenes.make_scene(cfg, host.generate_aabb)
    w, h = (sc.width, sc.height) if cfg <= 3 else ((1920, 1080))
    p = sc.params(width=w, height=h)
    rt.load(sc)
    rays = None
    ref = None
    for rounds in range(2):
        for v in variants:
            rt.set_variant(v)
            rt.render(p); rt.sync()
            ts = []
            for _ in range(5):
                rt.render(p); ts.append(rt.last_kernel_ms())
            col, pos, nrm = rt.readback()
            if rays is None: rays = rt.count_rays(p)
            if ref is None: ref = (col, pos, nrm)
            same = all(((a == b) | (np.isnan(a.astype(np.float32)) & np.isnan(b.astype(np.float32)))).all() for a, b in zip(ref, (col, pos, nrm)))
            rt.count_rays(p); st = rt.debug_stats()
            extra = f" packets {st[1]} cand/packet {st[2]/max(st[1],1):.2f} lanes/packet {st[0]/max(st[1],1):.1f}" if st[1] else ""
            print(f"cfg {cfg} {w}x{h} variant {v}: median {np.median(ts):.3f} ms  min {min(ts):.3f}  {rays/np.median(ts)/1e3:.0f} Mray/s  identical_to_v{variants[0]}={same} rays={rt.count_rays(p)==rays}{extra}", flush=True)
