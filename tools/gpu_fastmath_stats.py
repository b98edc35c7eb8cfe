"""How often do the IEEE fallbacks of csrc/rt_fastmath.h execute?  Builds the library with -DRT_FASTMATH_STATS=1 on the
GPU box, renders one frame per config and prints wave-level fallback executions per call site class.
usage: python tools/gpu_fastmath_stats.py [cfgs=2,3,4,5]"""
import ctypes, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from opengl_raytracing_amd import build as B
if not os.environ.get("RT_LIB"):
    out = "/tmp/lib_fmstats.so"
    B.build_library(force=True, verbose=False, extra_flags=("-DRT_FASTMATH_STATS=1",), out=out)
    sys.exit(subprocess.run([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=dict(os.environ, RT_LIB=out)).returncode)
from opengl_raytracing_amd import host, scenes
lib = host.load_library()
lib.rt_debug_fastmath_fallbacks.argtypes = [ctypes.c_void_p, ctypes.c_int]
rt = host.RayTracer(0)
for cfg in [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "2,3,4,5").split(",")]:
    sc = scenes.make_scene(cfg, host.generate_aabb)
    p = sc.params()
    rt.load(sc)
    rt.render(p); rt.sync()
    z = (ctypes.c_ulonglong * 4)()
    lib.rt_debug_fastmath_fallbacks(z, 1)
    rt.render(p); rt.sync()
    lib.rt_debug_fastmath_fallbacks(z, 1)
    waves = (sc.width // 8) * (sc.height // 8)
    print(f"C{cfg}: fallbacks per frame: rcp {z[0]}  rcp3 {z[1]}  sqrt {z[2]}  rcp_sqrt {z[3]}   ({waves} waves, {rt.count_rays(p)} rays)", flush=True)
