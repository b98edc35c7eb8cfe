"""Secondary measurement: the bloom chain (extract + 10 blur passes + combine) at 1080p / 4K on a ray-traced
frame.  Compulsory HBM traffic of the chain as launched (5 fused horizontal+vertical kernels): first pair
16 in + 8 out, three middle pairs 8 + 8, last pair 8 + 16 (scene) in + 16 out = 112 B/pixel (the unfused
12-kernel chain moved 224 B/pixel)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opengl_raytracing_amd import host, scenes

rt = host.RayTracer(0)
sc = scenes.make_scene(2, host.generate_aabb)
rt.load(sc)
s = torch.cuda.Stream()
for (w, h) in [(1920, 1080), (3840, 2160)]:
    p = sc.params(width=w, height=h)
    col = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    pos = torch.empty_like(col)
    nrm = torch.empty((h, w, 4), dtype=torch.float16, device="cuda")
    rt.render_to(p, col.data_ptr(), pos.data_ptr(), nrm.data_ptr(), stream=s.cuda_stream)
    out = torch.empty_like(col)
    for _ in range(3):
        rt.bloom(col.data_ptr(), out.data_ptr(), w, h, 1.0, 0.5, 10, stream=s.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 30
    e0.record(s)
    for _ in range(K):
        rt.bloom(col.data_ptr(), out.data_ptr(), w, h, 1.0, 0.5, 10, stream=s.cuda_stream)
    e1.record(s)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / K * 1e3
    gbs = w * h * 112 / (us * 1e-6) / 1e9
    print(json.dumps({"chain": "rt_bloom (5 fused kernels)", "size": [w, h], "us": round(us, 1), "algorithmic_GBps": round(gbs, 1),
                      "hbm_peak_GBps": 8000.0, "frac": round(gbs / 8000.0, 3), "bytes_per_px": 112}), flush=True)
