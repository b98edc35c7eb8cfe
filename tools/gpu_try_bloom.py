"""Build macro variants of the library on the GPU box and time the bloom chain with each (tools/bench_bloom.py), after the
bloom GPU tests.  usage: python tools/gpu_try_bloom.py "name:-DRT_BLOOM_TPW=3" ..."""
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
    t = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_bloom.py"), "-q", "-m", "gpu", "-x"], env=env, capture_output=True, text=True)
    print(f"[{name}] tests: {t.stdout.strip().splitlines()[-1] if t.stdout.strip() else t.stderr[-200:]}", flush=True)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "bench_bloom.py")], env=env, capture_output=True, text=True)
    for l in r.stdout.splitlines():
        if l.startswith("{"): print(f"[{name}] {l}", flush=True)
    if r.returncode: print(f"[{name}] FAILED {r.stderr[-300:]}", flush=True)
