"""How well do a frame's tile costs predict those of the frame 64 frameCounts later?  hammersley(depth*64 + frameCount, 64)
(raytracingCs.glsl:557) makes the bounce sample all pixels share nearly periodic in frameCount (period 64: phi exactly, cosTheta^2
within 2^-6), so the cost map of phase frameCount % 64 should repeat.  Per-frame cost maps (differences of the accumulated
rt_debug_tile_costs) for a few frameCounts, then a list-scheduling simulation (5 120 wave slots, next tile to the first free slot)
of frame B's costs in the order sorted from: its own costs (ideal), the frame 64 earlier, the frame 1 earlier, the mean of 32
other frames, raster."""
import heapq, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opengl_raytracing_amd import host, scenes, layout as L

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 5120
rt = host.RayTracer(0)
sc = scenes.make_scene(cfg, host.generate_aabb)
rt.load(sc)
rt.set_variant(1 | 0x200)          # measured costs only
base = sc.params()


def cost_of(fc):
    p = L.copy_params(base, frameCount=fc)
    rt.render(p); rt.sync()
    a = rt.tile_costs().astype(np.int64)
    rt.render(p); rt.sync()
    b = rt.tile_costs().astype(np.int64)
    return (b - a).ravel().astype(np.float64)


def makespan(cost, order):
    h = [0.0] * slots
    for t in order:
        heapq.heapreplace(h, h[0] + cost[t])
    return max(h)


fcs = [1, 2, 3, 65, 66, 129, 130, 17, 81]
maps = {f: cost_of(f) for f in fcs}
others = [cost_of(f) for f in range(200, 232)]
mean32 = np.mean(others, axis=0)
print(f"C{cfg}: {maps[1].size} tiles, {slots} slots; corr(1,65) {np.corrcoef(maps[1], maps[65])[0,1]:.3f}  corr(1,2) {np.corrcoef(maps[1], maps[2])[0,1]:.3f}  "
      f"corr(1,mean32) {np.corrcoef(maps[1], mean32)[0,1]:.3f}  corr(65,129) {np.corrcoef(maps[65], maps[129])[0,1]:.3f}")
for tgt, prev64, prev1 in [(65, 1, 3), (66, 2, 65), (129, 65, 66), (130, 66, 129), (81, 17, 3)]:
    c = maps[tgt]
    ideal = makespan(c, np.argsort(-c))
    res = {"own": ideal, "from -64": makespan(c, np.argsort(-maps[prev64])), "from another phase": makespan(c, np.argsort(-maps[prev1])),
           "mean of 32": makespan(c, np.argsort(-mean32)), "raster": makespan(c, np.arange(c.size))}
    lower = c.sum() / slots
    print(f"frameCount {tgt}: sum/slots {lower:.0f}, max tile {c.max():.0f} | " + "  ".join(f"{k} {v / lower:.3f}" for k, v in res.items()))
