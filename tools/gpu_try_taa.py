"""Build macro variants of the library on the GPU box and time the TAA resolve with each (tools/bench_taa.py), after the TAA
GPU tests.  usage: python tools/gpu_try_taa.py "name:-DRT_TAA_XCD=0" ..."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from opengl_raytracing_amd import build as B
os.makedirs("/tmp/rtx", exist_ok=True)
for spec in sys.argv[1:]:
    name, _, flags = spec.partition(":")
    out = f"/tmp/rtx/lib_{name}.so"
    B.build_library(force=True, verbose=False, extra_flags=tuple(flags.split()), out=out)
    env = dict(os.environ, RT_LIB=out)
    t = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_taa.py"), os.path.join(REPO, "tests", "test_frame.py"), "-q", "-m", "gpu", "-x"], env=env, capture_output=True, text=True)
    print(f"[{name}] tests: {t.stdout.strip().splitlines()[-1] if t.stdout.strip() else t.stderr[-200:]}", flush=True)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "bench_taa.py")], env=env, capture_output=True, text=True)
    for l in r.stdout.splitlines():
        if l.startswith("{"): print(f"[{name}] {l}", flush=True)
    if r.returncode: print(f"[{name}] FAILED {r.stderr[-300:]}", flush=True)
