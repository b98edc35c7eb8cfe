"""Section timers of the instrumented packet kernel (-DRT_PK_TIMERS=1, built on the GPU box): which share of the waves'
time goes to closest-hit traversal, a light's packet + candidate masks, its PCF sample loops, and everything else.
usage: python tools/gpu_timers.py [cfgs=4,5] [extra hipcc flags...]"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if os.environ.get("RT_TIMERS_CHILD"):
    from opengl_raytracing_amd import host, scenes
    rt = host.RayTracer(0)
    for cfg in [int(c) for c in sys.argv[1].split(",")]:
        sc = scenes.make_scene(cfg, host.generate_aabb)
        p = sc.params()
        rt.load(sc)
        rays = rt.count_rays_traced(p); s = rt.debug_stats_ex()
        tot = max(s[15], 1)
        print(f"C{cfg} {sc.width}x{sc.height}: traced rays {rays} packets {s[1]} cand/packet {s[2] / max(s[1], 1):.1f} | wave time: "
              f"closest {s[12] / tot:.3f}  light packet+masks {s[13] / tot:.3f}  pcf samples {s[14] / tot:.3f}  "
              f"rest {1 - (s[12] + s[13] + s[14]) / tot:.3f} | of the light section: set-up {s[16] / tot:.3f}, masks of split packets {s[17] / tot:.3f}; "
              f"split / unsplit light packets {s[18]} / {s[19]} | sample groups {s[25]}, candidate trips {s[22]} ({s[22] / max(s[25], 1):.1f} per group), "
              f"with a lane passing the slab test {s[23]} ({s[23] / max(s[22], 1):.3f}), lanes passing per such trip {s[24] / max(s[23], 1):.1f}; "
              f"wave clocks per sample group {s[14] / max(s[25], 1):.0f} | light loop {s[26] / tot:.3f} (of which PCSS blocker search {s[28] / tot:.3f}, "
              f"shading + light set-up {(s[26] - s[13] - s[14] - s[28]) / tot:.3f}), subsurface {s[27] / tot:.3f}, "
              f"hit shading set-up {s[29] / tot:.3f}, roulette + next direction {s[30] / tot:.3f}, "
              f"ray generation + sky + stores {1 - (s[12] + s[26] + s[27] + s[29] + s[30]) / tot:.3f}", flush=True)
    sys.exit(0)
from opengl_raytracing_amd import build as B
os.makedirs("/tmp/rtx", exist_ok=True)
cfgs = sys.argv[1] if len(sys.argv) > 1 else "4,5"
out = "/tmp/rtx/lib_timers.so"
if os.environ.get("RT_TIMERS_LIB"):      # prebuilt in the build container (tools/build_variants.py "timers:-DRT_PK_TIMERS=1")
    out = os.path.abspath(os.environ["RT_TIMERS_LIB"])
else:
    B.build_library(force=True, verbose=False, extra_flags=["-DRT_PK_TIMERS=1", *sys.argv[2:]], out=out)
r = subprocess.run([sys.executable, os.path.abspath(__file__), cfgs], env=dict(os.environ, RT_LIB=out, RT_TIMERS_CHILD="1"))
sys.exit(r.returncode)
