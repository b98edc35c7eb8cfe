"""Build several macro variants of the library ON the GPU box and time them in one gpurun call.
usage: python tools/gpu_explore.py "name1:-DX=1 -DY=2" "name2:..."  [--cfgs=2,3]"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from opengl_raytracing_amd import build as B
specs = [a for a in sys.argv[1:] if not a.startswith("--")]
cfgs = "2"
for a in sys.argv[1:]:
    if a.startswith("--cfgs="): cfgs = a.split("=")[1]
os.makedirs("/tmp/rtx", exist_ok=True)
for spec in specs:
    name, _, flags = spec.partition(":")
    out = f"/tmp/rtx/lib_{name}.so"
    import io, contextlib
    cmdflags = tuple(flags.split()) + ("-Rpass-analysis=kernel-resource-usage",)
    srcs = [os.path.join(B.CSRC, s_) for s_ in B.SOURCES]
    cmd = [B._hipcc(), *B.HIPCC_FLAGS, *cmdflags, "-I", os.path.join(REPO, "include"), "-I", B.CSRC, "-x", "hip", *srcs, "-o", out]
    cr = subprocess.run(cmd, capture_output=True, text=True)
    if cr.returncode: print(f"[{name}] BUILD FAILED {cr.stderr[-300:]}"); continue
    import re
    blocks = cr.stderr.split("Function Name: ")
    for b_ in blocks:
        if b_.startswith("_Z23rt_render_packet_kernelILi0") or b_.startswith("_Z16rt_render_kernelILi0"):
            g = lambda key: (re.search(key + r": (\d+)", b_) or [None, "?"])[1]
            vg, sc_, oc, sp = g("VGPRs"), g("ScratchSize .bytes.lane."), g("Occupancy .waves.SIMD."), g("SGPRs Spill")
            print(f"[{name}] {b_[:30]}: VGPR {vg} scratch {sc_} occ {oc} sgprspill {sp}", flush=True)
    env = dict(os.environ, RT_LIB=out)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "gpu_ab.py"), os.environ.get("RT_AB_VARIANTS", "1"), cfgs], env=env, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("cfg")]
    for l in lines[1::2]:
        print(f"[{name}] {l[:118]}", flush=True)
    if r.returncode: print(f"[{name}] FAILED\n{r.stderr[-400:]}")
