"""In-tree build of librt_mi355.so (hipcc, gfx950 only) and of the oracle (gcc).

``python -m opengl_raytracing_amd.build`` builds both; ``__graft_entry__.build()`` calls
``build_all()``.  The .so files stay in-tree (git-ignored) so they travel to the GPU box.
"""
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "librt_mi355.so")
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle_rt.so")

SOURCES = ["rt_kernels.hip", "rt_post.hip", "rt_abi.cpp", "rt_host.cpp", "rt_mgpu.cpp"]
HEADERS = [os.path.join(CSRC, "rt_device.h"), os.path.join(CSRC, "rt_packet.inc"), os.path.join(CSRC, "rt_shadowtab.inc"), os.path.join(CSRC, "rt_mesa_math.h"), os.path.join(CSRC, "rt_fastmath.h"), os.path.join(REPO, "include", "rt_mi355.h")]

# -ffp-contract=off: the reference's GL never fuses a*b+c; IEEE divide and
# sqrt are hipcc's default (-fhip-fp32-correctly-rounded-divide-sqrt).
# -fno-slp-vectorize: the SLP vectorizer pairs adjacent scalar fp32 operations into v_pk_mul_f32 / v_pk_add_f32; in these kernels
# the even-aligned register pairs and the moves that feed them cost more than the packed issue saves (measured, same run,
# bit-identical: C2 0.322 -> 0.312 ms, C4 4.65 -> 4.17, C5 24.8 -> 20.9, SSAO 385 -> 364 us; bloom's explicitly packed arithmetic
# and the TAA resolve unchanged).
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc", "-fno-slp-vectorize",
    "-Wall", "-Wextra", "-Wno-unused-parameter",
]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; this package has no CPU fallback)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, extra_flags=(), out=None):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    out = out or LIB_PATH
    if not force and not _stale(out, srcs + HEADERS + [os.path.abspath(__file__)]):
        return out
    cmd = [_hipcc(), *HIPCC_FLAGS, *extra_flags, "-I", os.path.join(REPO, "include"), "-I", CSRC,
           "-x", "hip", *srcs, "-o", out]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


def build_oracle(force=False, verbose=True):
    """gcc build of the CPU restatement (+ the llvmpipe harness when Mesa headers exist).
    Building the checker is not using it."""
    deps = [os.path.join(ORACLE_DIR, f) for f in ("rt_oracle.c", "rt_post_oracle.c", "rt_oracle.h", "Makefile")]
    if force or _stale(ORACLE_LIB, deps):
        if verbose:
            print("[build] make -C oracle", flush=True)
        subprocess.run(["make", "-C", ORACLE_DIR] + (["-B"] if force else []), check=True,
                       stdout=None if verbose else subprocess.DEVNULL)
    return ORACLE_LIB


def build_all(force=False, verbose=True):
    build_library(force=force, verbose=verbose)
    build_oracle(force=force, verbose=verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
