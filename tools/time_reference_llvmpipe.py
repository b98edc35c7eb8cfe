#!/usr/bin/env python3
"""Time the REFERENCE path itself -- /root/reference/shader/raytracingCs.glsl, unmodified, on Mesa llvmpipe through
oracle/_ref/gl_harness -- on this machine's host cores, and write profiles/reference_llvmpipe_timing.json, the static
record bench.py attaches as `cpu_baseline_reference` (north_star: "the reference path timed ... via Mesa llvmpipe";
/root/reference cannot travel to the GPU box, so this runs in the build container only).

Two dispatch shapes per config: exact ceil(W/32) x ceil(H/32), and the reference host's own as-shipped
(W+15)/16 x (H+15)/16 (ForwardShadingPipeline.cpp:175-179: 4x the invocations, the surplus ones run the whole
shader and have their imageStores discarded).  Median of `--repeat` dispatches after one warm-up (JIT).

    python tools/time_reference_llvmpipe.py [--configs 2] [--repeat 3]
"""
import argparse
import json
import os
import platform
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from opengl_raytracing_amd import host, scenes  # noqa: E402
from oracle import binding as O  # noqa: E402


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return platform.processor()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2")
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    if not O.harness_available():
        sys.exit("needs oracle/_ref/gl_harness and /root/reference (build container only)")
    path = os.path.join(REPO, "profiles", "reference_llvmpipe_timing.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    threads = len(os.sched_getaffinity(0))
    for cfg in [int(c) for c in args.configs.split(",")]:
        sc = scenes.make_scene(cfg, host.generate_aabb)
        p = sc.params()
        _, _, _, rays = O.render(sc, p)            # the unit count R (oracle == instrumented HIP kernel)
        rec = {"kind": "reference", "what": "raytracingCs.glsl unmodified on Mesa llvmpipe (oracle/_ref/gl_harness)",
               "script": "tools/time_reference_llvmpipe.py", "cores": threads, "LP_NUM_THREADS": threads,
               "cpu": cpu_model(), "machine": "build container (the GPU box has no /root/reference)",
               "workload": f"C{cfg}: {sc.width}x{sc.height}, {len(sc.objects)} objects, {len(sc.lights)} lights, depth {sc.max_ray_depth}",
               "rays_reference_per_frame": int(rays), "unit": "Mray/s"}
        for key, shipped in (("exact_dispatch", False), ("as_shipped_dispatch", True)):
            _, _, _, info = O.run_reference(sc, p, repeat=args.repeat, shipped_dispatch=shipped, threads=threads)
            s = info["median_dispatch_s"]
            rec[key] = {"groups": info["groups"], "s_per_frame": round(s, 4), "value": round(rays / s / 1e6, 3),
                        "first_dispatch_s_incl_jit": round(info["first_dispatch_s"], 4), "repeat": args.repeat}
            rec["renderer"] = info["renderer"] + " / " + info["version"]
            print(f"C{cfg} {key}: {s:.3f} s/frame = {rays / s / 1e6:.2f} Mray/s", flush=True)
        rec["value"] = rec["exact_dispatch"]["value"]
        out[f"c{cfg}"] = rec
        json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
