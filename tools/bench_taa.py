"""Secondary measurement (not the BASELINE metric): the TAA resolve pass at 1920x1080 and 3840x2160
on surfaces produced by the ray tracer, against the HBM roofline (56 B of compulsory traffic per pixel:
current 16 + history 16 + gNormal 8 read, 16 written)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opengl_raytracing_amd import host, scenes

rt = host.RayTracer(0)
sc = scenes.make_scene(2, host.generate_aabb)
rt.load(sc)
s = torch.cuda.Stream()
for (w, h) in [(1920, 1080), (3840, 2160), (7680, 4320)]:
    p = sc.params(width=w, height=h)
    col = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    pos = torch.empty_like(col)
    nrm = torch.empty((h, w, 4), dtype=torch.float16, device="cuda")
    rt.render_to(p, col.data_ptr(), pos.data_ptr(), nrm.data_ptr(), stream=s.cuda_stream)
    hist = [torch.zeros_like(col), torch.empty_like(col)]
    jx, jy = host.taa_jitter(5, w, h)
    for _ in range(5):
        rt.taa_resolve(col.data_ptr(), hist[0].data_ptr(), nrm.data_ptr(), hist[1].data_ptr(), w, h, 0.1, jx, jy, stream=s.cuda_stream)
        hist.reverse()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 50
    e0.record(s)
    for _ in range(K):
        rt.taa_resolve(col.data_ptr(), hist[0].data_ptr(), nrm.data_ptr(), hist[1].data_ptr(), w, h, 0.1, jx, jy, stream=s.cuda_stream)
        hist.reverse()
    e1.record(s)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / K * 1e3
    gbs = w * h * 56 / (us * 1e-6) / 1e9
    print(json.dumps({"kernel": "rt_taa_resolve", "size": [w, h], "us": round(us, 2), "algorithmic_GBps": round(gbs, 1),
                      "hbm_peak_GBps": 8000.0, "frac": round(gbs / 8000.0, 3), "bytes_per_px": 56}), flush=True)
