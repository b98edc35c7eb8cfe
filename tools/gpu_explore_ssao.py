"""Build macro variants of the library on the GPU box and time the SSAO kernel for each."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from opengl_raytracing_amd import build as B
os.makedirs("/tmp/rtx", exist_ok=True)
for spec in sys.argv[1:]:
    name, _, flags = spec.partition(":")
    out = f"/tmp/rtx/lib_{name}.so"
    B.build_library(force=True, verbose=False, extra_flags=tuple(flags.split()), out=out)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "bench_ssao.py")], env=dict(os.environ, RT_LIB=out), capture_output=True, text=True)
    for l in r.stdout.splitlines():
        if l.startswith("{"): print(f"[{name}] {l[:120]}", flush=True)
    t = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_ssao.py"), "-m", "gpu", "-x", "-q"], env=dict(os.environ, RT_LIB=out), capture_output=True, text=True)
    print(f"[{name}] tests: {t.stdout.strip().splitlines()[-1]}", flush=True)
