"""How good is the tile-cost predictor?  Predicted class against the MEASURED cost of every tile (rt_debug_tile_costs under the
measured-cost feedback mode), per config: rank correlation, the mean measured cost per class, and where the heaviest measured
tiles sit in the predicted order.  usage: python tools/gpu_predictor.py [cfgs=2,4]"""
import os, sys, heapq
os.environ["RT_DEBUG_PRED_CLASSES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opengl_raytracing_amd import host, scenes

for cfg in [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "2,4").split(",")]:
    sc = scenes.make_scene(cfg, host.generate_aabb)
    p = sc.params()
    rt = host.RayTracer(0)
    rt.load(sc)
    rt.set_variant(0x201)
    rt.render(p); rt.sync()              # (first frame of these inputs: its costs are stored, no sort clears them yet)
    cost = rt.tile_costs().astype(np.float64).ravel()
    rt.set_variant(1)
    rt.render(p); rt.sync()
    cls = rt.predicted_classes(cost.size).astype(np.int64)
    order_pred = np.argsort(-cls, kind="stable")
    rank = np.empty(cost.size, dtype=np.int64); rank[order_pred] = np.arange(cost.size)
    from scipy.stats import spearmanr
    rho = spearmanr(cls, cost).correlation
    heavy = np.argsort(-cost)[: cost.size // 20]
    print(f"C{cfg}: {cost.size} tiles, classes used {sorted(set(cls.tolist()))}, Spearman(class, measured) {rho:.3f}; the 5 % heaviest measured tiles "
          f"sit at predicted positions: median {np.median(rank[heavy]) / cost.size:.3f}, p90 {np.percentile(rank[heavy], 90) / cost.size:.3f}, max {rank[heavy].max() / cost.size:.3f} of the order")
    def makespan(order, slots=5120):
        h = [0.0] * slots
        heapq.heapify(h)
        end = 0.0
        for t in order:
            st = heapq.heappop(h); e = st + cost[t]; end = max(end, e); heapq.heappush(h, e)
        return end
    ideal = cost.sum() / 5120
    print(f"   list-scheduling model, 5120 slots, measured tile costs: ideal {ideal:.0f}; raster {makespan(np.arange(cost.size)) / ideal:.3f}x, "
          f"predicted classes {makespan(order_pred) / ideal:.3f}x, measured LPT {makespan(np.argsort(-cost)) / ideal:.3f}x of ideal")
    for c in sorted(set(cls.tolist()), reverse=True):
        m = cls == c
        print(f"   class {c:2d}: {int(m.sum()):6d} tiles, measured cost mean {cost[m].mean():9.0f}  p10 {np.percentile(cost[m], 10):9.0f}  p90 {np.percentile(cost[m], 90):9.0f}")
    os.makedirs("gpurun_out/r3", exist_ok=True)
    np.savez_compressed(f"gpurun_out/r3/pred_c{cfg}.npz", cost=cost.astype(np.float32), cls=cls.astype(np.uint8), tiles_x=(sc.width + 7) // 8)
    rt.close()
