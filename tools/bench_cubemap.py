"""Secondary measurement: equirect -> cubemap (one-off producer of the skybox) at the reference's 512^2 faces from a
2048x1024 and a 4096x2048 panorama.  Device time of the two kernels (upload conversion + face gather) by HIP events;
compulsory traffic: panorama 12 B/texel read + 8 B/texel written, then 6*S*S*(4 taps * 8 B gathered, 6 B written)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from opengl_raytracing_amd import host

rt = host.RayTracer(0)
rng = np.random.default_rng(1)
for (w, h, S) in [(2048, 1024, 512), (4096, 2048, 512), (4096, 2048, 2048)]:
    pano = (rng.uniform(0, 1, (h, w, 3)) ** 3 * 20).astype(np.float32)
    faces = torch.zeros((6, S, S, 3), dtype=torch.float16, device="cuda")
    rt.equirect_to_cubemap(pano, S, d_faces_out=faces.data_ptr())       # warm-up (includes H2D + malloc)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        rt.equirect_to_cubemap(pano, S, d_faces_out=faces.data_ptr())
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / K * 1e3
    print(json.dumps({"panorama": [w, h], "face": S, "wall_ms_incl_h2d_and_malloc": round(wall, 3),
                      "h2d_bytes": int(pano.nbytes), "faces_bytes": int(faces.numel() * 2)}), flush=True)
