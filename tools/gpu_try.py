"""Build macro variants of the library ON the GPU box and time them at the configs' full sizes in one gpurun call,
checking every variant's three surfaces against the first variant's bit for bit.
usage: python tools/gpu_try.py "base:" "name:-DRT_X=1 -DRT_Y=2" ... [--cfgs=2,4,5] [--reps=7] [--size=WxH] [--pcf=N] [--pcss=1]
A spec "name:@path/to/lib.so" takes a library prebuilt in the build container (tools/build_variants.py -> exp/) instead of
compiling on the box (box time is GPU budget)."""
import os, re, subprocess, sys, zlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from opengl_raytracing_amd import build as B

specs = [a for a in sys.argv[1:] if not a.startswith("--")]
opt = {a.split("=")[0][2:]: a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
cfgs, reps, size = opt.get("cfgs", "2"), int(opt.get("reps", "7")), opt.get("size")

if os.environ.get("RT_TRY_CHILD"):
    import numpy as np
    from opengl_raytracing_amd import host, scenes
    rt = host.RayTracer(0)
    for cfg in [int(c) for c in cfgs.split(",")]:
        sc = scenes.make_scene(cfg, host.generate_aabb)
        w, h = (int(v) for v in size.split("x")) if size else (sc.width, sc.height)
        p = sc.params(width=w, height=h)
        if "pcf" in opt:
            sc.lights["pcfSamples"] = int(opt["pcf"])
        if "pcss" in opt:                    # every light PCSS (shadowType 2): the PCSS kernel instantiations on any config's scene
            sc.lights["shadowType"] = 2
        rt.load(sc)
        for _ in range(3):
            rt.render(p); rt.sync()
        import time
        ts = []
        k = 40 if w * h <= 1920 * 1080 else (12 if w * h <= 3840 * 2160 else 5)
        for _ in range(reps):          # like bench.py: k back-to-back launches, wall clock around the batch
            rt.sync(); t0 = time.perf_counter()
            for _ in range(k):
                rt.render(p)
            rt.sync(); ts.append((time.perf_counter() - t0) / k * 1e3)
        col, pos, nrm = rt.readback()
        crc = zlib.crc32(nrm.tobytes(), zlib.crc32(pos.tobytes(), zlib.crc32(col.tobytes())))
        from opengl_raytracing_amd import layout as L
        # frames whose inputs differ from each other (frameCount advances: the reference's default, TAA on) -- the SAME sequence
        # under the default scheduler (order predicted per frame) and in raster order; back to back, and one frame at a time
        seq = [L.copy_params(p, frameCount=sc.frame_count + 1 + i) for i in range(max(k, 12))]
        res = {}
        for mode, name in ((1, "auto"), (0x101, "raster")):
            rt.set_variant(mode)
            for q in seq[:4]:
                rt.render(q)
            rt.sync(); t0 = time.perf_counter()
            for q in seq:
                rt.render(q)
            rt.sync(); res[name] = (time.perf_counter() - t0) / len(seq) * 1e3
            lat = []
            for q in seq[:10]:
                rt.sync(); t0 = time.perf_counter(); rt.render(q); rt.sync(); lat.append((time.perf_counter() - t0) * 1e3)
            res[name + "_alone"] = float(np.mean(lat))
        rt.set_variant(0x101)
        for _ in range(3):
            rt.render(p)
        rt.sync(); t0 = time.perf_counter()
        for _ in range(k):
            rt.render(p)
        rt.sync(); raster = (time.perf_counter() - t0) / k * 1e3
        rt.set_variant(1)
        print(f"RESULT C{cfg} {w}x{h}: static {np.median(ts):.4f} ms (raster {raster:.4f}) | new frames: predicted {res['auto']:.4f}  raster {res['raster']:.4f} | "
              f"one at a time: predicted {res['auto_alone']:.4f}  raster {res['raster_alone']:.4f}  crc {crc:08x}", flush=True)
    sys.exit(0)

os.makedirs("/tmp/rtx", exist_ok=True)
ref = {}
for spec in specs:
    name, _, flags = spec.partition(":")
    out = f"/tmp/rtx/lib_{name}.so"
    if flags.startswith("@"):
        out = os.path.abspath(flags[1:])
        if not os.path.exists(out):
            print(f"[{name}] missing {out}", flush=True); continue
    srcs = [os.path.join(B.CSRC, s_) for s_ in B.SOURCES]
    cmd = [B._hipcc(), *B.HIPCC_FLAGS, *flags.split(), "-Rpass-analysis=kernel-resource-usage", "-I", os.path.join(REPO, "include"),
           "-I", B.CSRC, "-x", "hip", *srcs, "-o", out]
    cr = subprocess.run(cmd, capture_output=True, text=True) if not flags.startswith("@") else subprocess.CompletedProcess(cmd, 0, "", "")
    if cr.returncode:
        print(f"[{name}] BUILD FAILED {cr.stderr[-600:]}", flush=True); continue
    for b_ in cr.stderr.split("Function Name: "):
        m = re.match(r"_Z23rt_render_packet_kernelILi0ELi(\d+)ELb(\d)E(\S*?)(Pk\w+?)E", b_)
        if m:
            g = lambda key: (re.search(key + r": (\d+)", b_) or [None, "?"])[1]
            print(f"[{name}] kernel<0,{m.group(1)},{m.group(4)}>: VGPR {g('VGPRs')} scratch {g('ScratchSize .bytes.lane.')} "
                  f"occ {g('Occupancy .waves.SIMD.')} sgpr-spill {g('SGPRs Spill')} vgpr-spill {g('VGPRs Spill')}", flush=True)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), *[a for a in sys.argv[1:] if a.startswith("--")]],
                       env=dict(os.environ, RT_LIB=out, RT_TRY_CHILD="1"), capture_output=True, text=True)
    for l in r.stdout.splitlines():
        if l.startswith("RESULT"):
            key = l.split(":")[0]
            crc = l.rsplit("crc ", 1)[1]
            ref.setdefault(key, crc)
            print(f"[{name}] {l[7:]}  identical_to_first={crc == ref[key]}", flush=True)
    if r.returncode:
        print(f"[{name}] FAILED\n{r.stderr[-600:]}", flush=True)
