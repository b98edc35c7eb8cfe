"""Per-rank render time of the strip plan on ONE GPU: what each rank of an N-GPU run would spend in the
render kernel per frame (the critical path of the strong-scaling bench), back-to-back on one stream and
with consecutive frames alternating between two streams (frames k and k+1 overlapping on the device)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opengl_raytracing_amd import dist as D
from opengl_raytracing_amd import host, scenes

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rt = host.RayTracer(0)
sc = scenes.make_scene(cfg, host.generate_aabb)
rt.load(sc)
W, H = sc.width, sc.height
base = sc.params()
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
K = 200
# clocks: a few hundred frames before anything is timed
_p0 = sc.params()
for _ in range(400):
    rt.render(_p0)
rt.sync()
for world in (1, 2, 4, 8):
    strip = D.default_strip_rows(H, world, len(sc.objects))
    plan = D.StripPlan(W, H, strip, world) if world > 1 else D.StripPlan(W, H, H, 1)
    res = {"world": world, "strip_rows": strip}
    for rank in sorted({0, world - 1}):
        p = plan.params(base, rank) if world > 1 else base
        bufs = [D.alloc_rank_buffer(plan, dev) for _ in range(4)]
        views = [D.surface_views(b, plan) for b in bufs]
        for mode in ("one_stream", "raster_order", "two_streams", "three_streams", "four_streams"):
            rt.set_variant(0x101 if mode == "raster_order" else 1)
            nst = {"two_streams": 2, "three_streams": 3, "four_streams": 4}.get(mode, 1)
            def frame(k):
                s = streams[k % nst]
                c, q, n = views[k % nst]
                rt.render_to(p, c.data_ptr(), q.data_ptr(), n.data_ptr(), stream=s.cuda_stream)
            for k in range(40):
                frame(k)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(streams[0])
            for st in streams[1:]:
                st.wait_event(e0)
            for k in range(K):
                frame(k)
            for st in streams[1:]:
                streams[0].wait_stream(st)
            e1.record(streams[0])
            torch.cuda.synchronize()
            res[f"rank{rank}_{mode}_ms"] = round(e0.elapsed_time(e1) / K, 4)
    print(json.dumps(res), flush=True)
