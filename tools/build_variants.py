"""Build macro variants of the library HERE (build container, no GPU) into exp/lib_<name>.so, which travel to the GPU box with
the snapshot; tools/gpu_try.py "name:@exp/lib_name.so" times them there.
usage: python tools/build_variants.py "name:-DRT_X=1 -DRT_Y=2" ...   (prints each packet kernel's register / scratch figures)"""
import os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from opengl_raytracing_amd import build as B

os.makedirs(os.path.join(REPO, "exp"), exist_ok=True)


def one(spec):
    name, _, flags = spec.partition(":")
    out = os.path.join(REPO, "exp", f"lib_{name}.so")
    srcs = [os.path.join(B.CSRC, s_) for s_ in B.SOURCES]
    cmd = [B._hipcc(), *B.HIPCC_FLAGS, *flags.split(), "-Rpass-analysis=kernel-resource-usage", "-I", os.path.join(REPO, "include"),
           "-I", B.CSRC, "-x", "hip", *srcs, "-o", out]
    cr = subprocess.run(cmd, capture_output=True, text=True)
    lines = []
    if cr.returncode:
        return f"[{name}] BUILD FAILED\n{cr.stderr[-1500:]}"
    for b_ in cr.stderr.split("Function Name: "):
        m = re.match(r"_Z23rt_render_packet_kernelILi0ELi(\d+)ELb(\d)E(\S*?)(Pk\w+?)E", b_)
        if m:
            g = lambda key: (re.search(key + r": (\d+)", b_) or [None, "?"])[1]
            lines.append(f"[{name}] kernel<0,{m.group(4)}>: VGPR {g('VGPRs')} scratch {g('ScratchSize .bytes.lane.')} "
                         f"occ {g('Occupancy .waves.SIMD.')} sgpr-spill {g('SGPRs Spill')} vgpr-spill {g('VGPRs Spill')}")
    return "\n".join(lines)


with ThreadPoolExecutor(max_workers=int(os.environ.get("RT_BUILD_JOBS", "4"))) as ex:
    for r in ex.map(one, [a for a in sys.argv[1:] if not a.startswith("--")]):
        print(r, flush=True)
