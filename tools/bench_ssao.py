"""Secondary measurement: SSAO + its blur pass on a ray-traced G-buffer at 1080p / 4K.  Per pixel: 64 samples x
(two mat4*vec4, 3 IEEE divides, one 4-byte depth gather); compulsory HBM traffic 16 (gPosition) + 8 (gNormal) read
+ 4 written = 28 B/pixel -- the kernel is VALU/gather-bound, not HBM-bound."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opengl_raytracing_amd import host, scenes

rt = host.RayTracer(0)
sc = scenes.make_scene(2, host.generate_aabb)
rt.load(sc)
s = torch.cuda.Stream()
samples, noise = host.ssao_kernel()
for (w, h) in [(1920, 1080), (3840, 2160)]:
    p = sc.params(width=w, height=h)
    col = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    pos = torch.empty_like(col)
    nrm = torch.empty((h, w, 4), dtype=torch.float16, device="cuda")
    rt.render_to(p, col.data_ptr(), pos.data_ptr(), nrm.data_ptr(), stream=s.cuda_stream)
    view, proj = host.camera_matrices(p.camPos[:], p.camDir[:], p.camUp[:], p.fovDeg, w / h)
    ao = torch.empty((h, w), dtype=torch.float32, device="cuda")
    ab = torch.empty_like(ao)
    res = {"size": [w, h]}
    for name, fn in (("rt_ssao", lambda: rt.ssao(pos.data_ptr(), nrm.data_ptr(), ao.data_ptr(), w, h, noise, samples, proj, view, stream=s.cuda_stream)),
                     ("rt_ssao_blur", lambda: rt.ssao_blur(ao.data_ptr(), ab.data_ptr(), w, h, False, stream=s.cuda_stream))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = 30
        e0.record(s)
        for _ in range(K):
            fn()
        e1.record(s)
        torch.cuda.synchronize()
        res[name + "_us"] = round(e0.elapsed_time(e1) / K * 1e3, 1)
    res["ssao_Gsamples_per_s"] = round(w * h * 64 / (res["rt_ssao_us"] * 1e-6) / 1e9, 1)
    res["ssao_compulsory_GBps"] = round(w * h * 28 / (res["rt_ssao_us"] * 1e-6) / 1e9, 1)
    res["blur_GBps"] = round(w * h * 8 / (res["rt_ssao_blur_us"] * 1e-6) / 1e9, 1)
    print(json.dumps(res), flush=True)
