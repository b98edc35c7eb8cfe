"""Distribution of per-tile (= per-wave for the 8x8-tile kernel) costs of the bench frame: the longest tile
is the critical path of a strip-parallel frame once every tile is resident at once (N >= 4 GPUs at 1080p)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opengl_raytracing_amd import host, scenes

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rt = host.RayTracer(0)
sc = scenes.make_scene(cfg, host.generate_aabb)
rt.load(sc)
p = sc.params()
for _ in range(50):
    rt.render(p)
rt.sync()
c = rt.tile_costs().astype(np.float64) * 64 / 2.4e3      # -> microseconds at 2.4 GHz
flat = np.sort(c.ravel())[::-1]
print(json.dumps({"tiles": int(flat.size), "grid": list(c.shape), "kernel_ms": rt.last_kernel_ms(),
                  "us_max": round(flat[0], 1), "us_top10": [round(x, 1) for x in flat[:10]],
                  "us_p99": round(float(np.percentile(flat, 99)), 1), "us_p90": round(float(np.percentile(flat, 90)), 1),
                  "us_median": round(float(np.median(flat)), 1), "us_mean": round(float(flat.mean()), 1),
                  "sum_ms_over_4096_slots": round(float(flat.sum()) / 4096 / 1e3, 4)}))
ys, xs = np.unravel_index(np.argsort(c.ravel())[::-1][:12], c.shape)
print("heaviest tiles (tileY, tileX, us):", [(int(y), int(x), round(float(c[y, x]), 1)) for y, x in zip(ys, xs)])
# coarse cost map, 8x8 tiles per cell (max within the cell)
H, W = c.shape
for y in range(H - 1, -1, -8):
    print(" ".join(f"{int(c[max(y - 7, 0):y + 1, x:x + 8].max()):4d}" for x in range(0, W, 8)))
