"""Build macro variants of the library on the GPU box and run bench.py (full-size configs) with each.
usage: python tools/gpu_explore_bench.py "name:-DX=1" ... --cfgs=4,5 [--steps=4]"""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from opengl_raytracing_amd import build as B
specs = [a for a in sys.argv[1:] if not a.startswith("--")]
cfgs, steps = "4,5", "4"
for a in sys.argv[1:]:
    if a.startswith("--cfgs="): cfgs = a.split("=")[1]
    if a.startswith("--steps="): steps = a.split("=")[1]
os.makedirs("/tmp/rtx", exist_ok=True)
for spec in specs:
    name, _, flags = spec.partition(":")
    out = f"/tmp/rtx/lib_{name}.so"
    B.build_library(force=True, verbose=False, extra_flags=tuple(flags.split()), out=out)
    for cfg in cfgs.split(","):
        r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--config", cfg, "--steps", steps, "--warmup", "2", "--no-cpu-baseline"],
                           env=dict(os.environ, RT_LIB=out), capture_output=True, text=True)
        for l in r.stdout.splitlines():
            if l.startswith("{"):
                d = json.loads(l)
                print(f"[{name}] C{cfg} {d['config']['width']}x{d['config']['height']}: {d['ms_per_step']:.3f} ms/frame  {d['value']:.0f} Mray/s", flush=True)
        if r.returncode: print(f"[{name}] C{cfg} FAILED {r.stderr[-300:]}", flush=True)
