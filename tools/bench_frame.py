"""Secondary measurement: rt_frame = the whole GPU side of one ForwardShadingPipline::Render() iteration (ray trace,
SSAO + blur, 12-pass bloom, TAA resolve) on the bench workload, per frame, with and without the (upstream-dead) AO."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opengl_raytracing_amd import host, scenes

rt = host.RayTracer(0)
sc = scenes.make_scene(2, host.generate_aabb)
rt.load(sc)
samples, noise = host.ssao_kernel()
for (w, h) in [(1920, 1080), (3840, 2160)]:
    p = sc.params(width=w, height=h)
    disp = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    res = {"size": [w, h]}
    for label, ao in (("ray+bloom+taa", False), ("ray+ao+bloom+taa", True)):
        def frame(k):
            p.frameCount = k
            rt.frame(p, enable_ao=ao, enable_taa=True, ao_samples=samples, ao_noise=noise, d_display=disp.data_ptr())
        for k in range(30):
            frame(k)
        rt.sync()
        K = 100
        t0 = time.perf_counter()
        for k in range(K):
            frame(30 + k)
        rt.sync()
        res[label + "_ms"] = round((time.perf_counter() - t0) / K * 1e3, 4)
    print(json.dumps(res), flush=True)
