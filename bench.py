#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its named workload, on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C] [--extra-configs 3,4,5|none]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one frame: one pass of the ray-tracing hot path over the workload's pixels, with the scene, textures
and output surfaces already resident in HBM (the reference re-uploads its <= 4 KB of SSBOs per frame; the
PCIe-inclusive figure is in DESIGN.md, never `value`).

Headline workload (N = 1 and N > 1 alike): BASELINE.json configs[1] = C2 -- 1920x1080, 16 spheres + 2 planes,
3 lights (point / directional / area), MAX_RAY_DEPTH 4, PCF x4 shadows, synthetic scene of SURVEY.md 8(d).  N > 1
splits that one frame into interleaved row strips, one process per GPU, and assembles the image on rank 0 with one
RCCL gather per frame + a copy kernel -- all inside the timed step ("scaling": "strong").

Rank 0 prints ONE JSON line.  What the fields mean (VERDICT r1 asked for a record that can be falsified):

* `value` = REFERENCE rays per second: rays = intersectObjects calls the reference shader makes for this frame
  (`rays_reference_per_frame`, counted exactly by an instrumented launch and equal to the CPU oracle's count) --
  the unit the CPU baselines are measured in.  `rays_traced_per_frame` is what the timed kernel really traverses
  (it skips rays whose result provably cannot reach a pixel); `mray_s_traced` uses that.
* `roofline` is the BINDING roofline, frac <= 1: VALU issue.  achieved = wave-level VALU instructions of the
  dominant kernel per launch (SQ_INSTS_VALU from the committed rocprofv3 PMC pass named in `pmc_source`; a
  deterministic property of kernel + scene) / the kernel's live HIP-event duration; peak = 1024 SIMDs x one wave64
  VALU instruction per 2 cycles x 2.4 GHz.  `traffic` = physical HBM bytes per launch from the same PMC passes
  (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md's gfx950 correction).
* `hbm_physical` = those bytes / the live kernel time against the 8 TB/s peak (the kernel is NOT HBM-bound: the scene
  lives in LDS / SGPRs); `wasted_traffic_ratio` = traffic / compulsory bytes (surfaces + inputs).
* `algorithmic_equiv` = SURVEY.md 8(d)'s algorithmic bytes (the 176 B/object/ray stream the reference shader
  reads) / kernel time: an EQUIVALENT rate with no fraction -- that stream is served on chip by design.
* `frame_ms` = per-frame distribution of the timed region (every frame bracketed by its own HIP events on the launch
  stream): median / p10 / p90 / min / max.  `ms_per_step` stays the mean the driver checks; `step_ms_device` is the
  same region by its first and last event, and the duration `roofline.frac` is computed from.
* The headline frames are identical, so from the third on they run in the order sorted from their MEASURED tile costs.
  `free_running` = the same workload with frameCount advancing every frame (the reference's default: TAA on,
  ForwardShadingPipeline.cpp:254): every frame's inputs are new.  frameCount only rotates the bounce sample all pixels
  share, with period 64 (raytracingCs.glsl:557), so the scheduler keeps one measured tile order per phase: `free_running`
  is the steady state (second period on: each frame runs in the order measured 64 frames earlier),
  `free_running_first_period` the first 64 new frames (all-phase average order).  Frames of >= 49 152 tiles without any
  measured order run in an order PREDICTED from their own inputs (rt_predict_tiles_kernel).  `cold_frame_ms` = new
  steady-state frames one at a time with a device sync in between (nothing overlaps; host launch + sync included);
  `raster_order_ms` = the scheduler switched off.  `scene_update_every_frame_ms` = the headline frame with rt_set_scene of
  CHANGED bytes in front of every frame (scene compile + shadow-table build; identical bytes, which is
  what the reference re-uploads while nothing is edited, are a no-op).
* `pipelined` = the headline frames with two in flight on two streams (the ABI is asynchronous): frame k's tail
  overlaps frame k+1's head.  Throughput, not latency.
* `cpu_baseline` = the oracle port timed on this box's host cores; `cpu_baseline_reference` = the reference's own
  GLSL on Mesa llvmpipe (static record produced by tools/time_reference_llvmpipe.py in the build container --
  /root/reference cannot travel to the GPU box).
* `configs_extra` = the same measurement for C3, C4, C5 (fewer steps), so they stop being prose.
* N > 1 only, `peer_store` = the same frame on the same N devices from ONE process (rt_mgpu_*, csrc/rt_mgpu.cpp): every
  device's kernel stores its strips straight into device 0's frame over xGMI, no wire format, no gather, no re-assembly.
  Measured by a child process of rank 0 after the ranks' own measurement (they wait on the rendezvous store, off the GPUs),
  under a timeout; reported BESIDE `value`, never instead of it (the contract's launch is one process per GPU).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s HBM3E peak
FP32_VALU_PEAK_TFLOPS = 157.3
N_SIMD = 256 * 4                      # 256 CUs x 4 SIMD-32
CLOCK_HZ = 2.4e9                      # max clock; the chip may run lower under load, so utilisation is a lower bound
VALU_ISSUE_PEAK = N_SIMD * CLOCK_HZ / 2.0 / 1e9      # G wave64 VALU instructions / s (2 cycles each on a SIMD-32)


def algorithmic_bytes(rays, n_obj, n_lt, n_px, noise_bound, sky_taps):
    """the scene-record stream the reference shader reads per ray
    (`Object obj = objects[i]`, raytracingCs.glsl:159-160) + compulsory outputs/inputs."""
    return rays * n_obj * 176 + n_px * 40 + (n_px if noise_bound else 0) + sky_taps * 24 + n_obj * 176 + n_lt * 96


def host_threads():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def kernel_source_hash():
    """sha256 over the kernel sources: ties a committed PMC record to the code it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(REPO, "opengl_raytracing_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".inc", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load_counters(cfg):
    """profiles/kernel_counters.json[cN] (written by profiles/summarize.py on the GPU box) or None."""
    path = os.path.join(REPO, "profiles", "kernel_counters.json")
    try:
        return json.load(open(path)).get(f"c{cfg}")
    except Exception:
        return None


def workload_name(cfg, sc):
    import numpy as _np
    st = int(sc.lights["shadowType"][0]) if len(sc.lights) else 0
    return (f"C{cfg}: {sc.width}x{sc.height}, {len(sc.objects)} objects ({int((sc.objects['type'] == 0).sum())} spheres + "
            f"{int((sc.objects['type'] == 1).sum())} planes), {len(sc.lights)} lights, depth {sc.max_ray_depth}, "
            f"{'PCSS' if st == 2 else 'PCF x4'} shadows" + (", noise texture" if sc.noise is not None else "") +
            (", skybox" if sc.use_skybox else ""))


def roofline_record(cfg, sc, n_px, rays_ref, step_ms, counters, src_hash):
    """The roofline / traffic objects of one config from the live duration of a step (HIP events over the timed region) and the
    committed PMC record."""
    ks = step_ms * 1e-3
    n_obj, n_lt = len(sc.objects), len(sc.lights)
    compulsory = n_px * 40 + n_obj * 176 + n_lt * 96 + (n_px if sc.noise is not None else 0)
    b_alg = algorithmic_bytes(rays_ref, n_obj, n_lt, n_px, sc.noise is not None, 0)
    out = {}
    traffic = None
    if counters:
        stale = counters.get("src_hash") != src_hash          # a record without a hash is NOT tied to this build
        insts = counters.get("SQ_INSTS_VALU")
        if counters.get("fetch_size_kb") is not None and counters.get("write_size_kb") is not None:
            traffic = int((2 * counters["fetch_size_kb"] + counters["write_size_kb"]) * 1024)
        if insts:
            achieved = insts / ks / 1e9
            out["roofline"] = {
                "bound": "valu-issue", "achieved": round(achieved, 1), "peak": round(VALU_ISSUE_PEAK, 1),
                "unit": "G wave64 VALU instr/s", "frac": round(achieved / VALU_ISSUE_PEAK, 3), "traffic": traffic,
                "valu_instructions_per_launch": int(insts),
                "pmc_source": counters.get("source"), "pmc_kernel": counters.get("kernel"),
                "pmc_kernel_avg_us": counters.get("kernel_avg_us"), "pmc_matches_this_build": (not stale),
                "salu_per_valu": (round(counters["SQ_INSTS_SALU"] / insts, 3) if counters.get("SQ_INSTS_SALU") else None),
                "note": "binding roofline: wave64 VALU issue slots (1024 SIMD-32 x 1 instr / 2 cycles x 2.4 GHz); instruction "
                        "count from the committed PMC pass (deterministic per kernel + scene), duration live (HIP events)"}
    if "roofline" not in out:
        out["roofline"] = {"bound": "valu-issue", "achieved": None, "peak": round(VALU_ISSUE_PEAK, 1),
                           "unit": "G wave64 VALU instr/s", "frac": None, "traffic": traffic,
                           "note": "no committed PMC record for this config (profiles/kernel_counters.json)"}
    if traffic is not None:
        out["hbm_physical"] = {"bytes": traffic, "GBps": round(traffic / ks / 1e9, 1),
                               "frac": round(traffic / ks / 1e9 / HBM_PEAK_GBS, 4), "peak_GBps": HBM_PEAK_GBS}
        out["wasted_traffic_ratio"] = round(traffic / compulsory, 3)
    out["compulsory_bytes"] = compulsory
    out["algorithmic_equiv"] = {
        "bytes_per_launch": b_alg, "GBps_equivalent": round(b_alg / ks / 1e9, 1),
        "note": "SURVEY.md 8(d): rays_reference*nObj*176 + px*40 + scene; served from LDS / SGPRs by design -- an "
                "equivalent rate, NOT physical bandwidth, hence no fraction of any peak"}
    return out


def measure_single(cfg, steps, warmup, variant, dev_index, with_modes=True):
    """One config on one GPU: fixed-frameCount headline, kernel-only duration, free-running and cold modes."""
    import torch
    from opengl_raytracing_amd import dist as D
    from opengl_raytracing_amd import host, scenes, layout as L

    dev = torch.device("cuda", dev_index)
    sc = scenes.make_scene(cfg, host.generate_aabb)
    W, H = sc.width, sc.height
    base = sc.params()
    rt = host.RayTracer(dev_index)
    rt.load(sc)
    rt.set_variant(variant)
    stream = torch.cuda.Stream(device=dev)
    assert stream.cuda_stream != 0
    one = D.StripPlan(W, H, H, 1)
    buf = D.alloc_rank_buffer(one, dev)
    col, pos, nrm = D.surface_views(buf, one)

    def render(p):
        rt.render_to(p, col.data_ptr(), pos.data_ptr(), nrm.data_ptr(), stream=stream.cuda_stream)

    rays_ref = rt.count_rays(base)
    rays_traced = rt.count_rays_traced(base)
    for _ in range(warmup):
        render(base)
    torch.cuda.synchronize()
    # the timed region: K frames, each between two events on the launch stream (K + 1 events)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    for i in range(steps):
        evs[i].record(stream)
        render(base)
    evs[steps].record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_frame = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)])
    res = dict(sc=sc, W=W, H=H, rays_ref=rays_ref, rays_traced=rays_traced, elapsed=elapsed,
               step_ms_dev=evs[0].elapsed_time(evs[steps]) / steps,
               frame_ms={"median": round(float(np.median(per_frame)), 4), "p10": round(float(np.percentile(per_frame, 10)), 4),
                         "p90": round(float(np.percentile(per_frame, 90)), 4), "min": round(float(per_frame.min()), 4),
                         "max": round(float(per_frame.max()), 4)})
    if with_modes:
        # (1) frameCount advancing every frame, as in the reference with TAA on (ForwardShadingPipeline.cpp:254).  Every frame's inputs
        #     are new and its rays differ, so each timed frame is counted.  The bounce sample all pixels share is (nearly) periodic in
        #     frameCount with period 64, and the scheduler keeps one measured tile order per phase (rt_abi.cpp): the FIRST period runs
        #     in the all-phase average order, later ones in the order measured 64 frames earlier.  Both are reported.
        n_free = min(steps, 64)
        fc0 = sc.frame_count
        frames = [L.copy_params(base, frameCount=fc0 + 1 + k) for k in range(8 + 64 + n_free + 16)]

        def timed(fr):
            rays = [rt.count_rays(p) for p in fr]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for p in fr:
                render(p)
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
            return {"ms_per_step": round(t / len(fr) * 1e3, 4), "frames": len(fr), "value_mray_s": round(sum(rays) / t / 1e6, 1),
                    "rays_reference_per_frame_min_max": [int(min(rays)), int(max(rays))]}
        for p in frames[:8]:
            render(p)
        first = timed(frames[8:8 + n_free])
        for p in frames[8 + n_free:72]:          # the rest of the first period, untimed
            render(p)
        res["free_running"] = timed(frames[72:72 + n_free])
        res["free_running"]["note"] = ("frameCount advances every frame (reference default, TAA on: ForwardShadingPipeline.cpp:254): every frame's "
                                       "inputs are new; steady state = from the second 64-frame period of frameCount on (about one second of "
                                       "the application), when each frame runs in the tile order measured on the frame 64 earlier")
        first["note"] = "the first period of a free-running frameCount: no phase has been seen yet, tiles run in the all-phase average order"
        res["free_running_first_period"] = first
        # (2) new frames one at a time (device sync in between, nothing overlaps): what a new frame costs in that steady state
        n_cold = max(4, min(steps, 16))
        cold = frames[72 + n_free:72 + n_free + n_cold]
        cold_rays = [rt.count_rays(p) for p in cold]
        lat = []
        for p in cold:
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            render(p)
            stream.synchronize()                 # what the frame's consumer waits for (the scheduler's sort runs beside, on its own stream)
            lat.append((time.perf_counter() - c0) * 1e3)
        res["cold_frame_ms"] = round(float(np.mean(lat)), 4)
        res["cold_frame_rays_reference_mean"] = int(np.mean(cold_rays))
        # (2b) the scene CHANGES every frame (an object dragged in the editor): rt_set_scene with new bytes -- staging copy, scene
        #      compile and the shadow-table build -- in front of every frame (on the context's stream,
        #      ordered behind the previous frame and in front of the next by events)
        moved = scenes.make_scene(cfg, host.generate_aabb)
        moved.objects["position"][0, 0] += 0.125
        host.generate_aabb(moved.objects)
        pair = [moved, sc]
        n_dyn = max(8, min(steps, 32))
        for k in range(4):
            rt.set_scene(pair[k & 1].objects, pair[k & 1].lights)
            render(base)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n_dyn):
            rt.set_scene(pair[k & 1].objects, pair[k & 1].lights)
            render(base)
        torch.cuda.synchronize()
        res["scene_update_every_frame_ms"] = round((time.perf_counter() - t0) / n_dyn * 1e3, 4)
        rt.set_scene(sc.objects, sc.lights)
        # (3) scheduler off: raster tile order, the headline frame
        rt.set_variant((variant & 0xff) | 0x100)
        for _ in range(2):
            render(base)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_cold):
            render(base)
        torch.cuda.synchronize()
        res["raster_order_ms"] = round((time.perf_counter() - t0) / n_cold * 1e3, 4)
        rt.set_variant(variant)
        # (4) two frames in flight on two streams (headline frames)
        s2 = torch.cuda.Stream(device=dev)
        buf2 = D.alloc_rank_buffer(one, dev)
        col2, pos2, nrm2 = D.surface_views(buf2, one)
        targets = [(stream, (col, pos, nrm)), (s2, (col2, pos2, nrm2))]

        def render2(k):
            st, (c_, q_, n_) = targets[k & 1]
            rt.render_to(base, c_.data_ptr(), q_.data_ptr(), n_.data_ptr(), stream=st.cuda_stream)
        for k in range(max(8, warmup)):
            render2(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            render2(k)
        torch.cuda.synchronize()
        t_pipe = time.perf_counter() - t0
        res["pipelined"] = {"frames_in_flight": 2, "ms_per_step": round(t_pipe / steps * 1e3, 4),
                            "value_mray_s": round(rays_ref * steps / t_pipe / 1e6, 1),
                            "note": "the same frames, two in flight on two streams: throughput, not latency"}
    rt.close()
    return res


def single_record(cfg, r, steps, warmup, src_hash):
    sc, n_px = r["sc"], r["W"] * r["H"]
    ms = r["elapsed"] / steps * 1e3
    out = {"workload": workload_name(cfg, sc), "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 4),
           "value_mray_s": round(r["rays_ref"] * steps / r["elapsed"] / 1e6, 1),
           "mray_s_traced": round(r["rays_traced"] * steps / r["elapsed"] / 1e6, 1),
           "mpx_per_s": round(n_px * steps / r["elapsed"] / 1e6, 1),
           "rays_reference_per_frame": r["rays_ref"], "rays_traced_per_frame": r["rays_traced"],
           "step_ms_device": round(r["step_ms_dev"], 4), "frame_ms": r["frame_ms"]}
    # the roofline's duration is the timed region's own (the one `value` is computed from), by HIP events on the launch stream
    out.update(roofline_record(cfg, sc, n_px, r["rays_ref"], r["step_ms_dev"], load_counters(cfg), src_hash))
    for k in ("free_running", "free_running_first_period", "cold_frame_ms", "cold_frame_rays_reference_mean", "scene_update_every_frame_ms",
              "raster_order_ms", "pipelined"):
        if k in r:
            out[k] = r[k]
    return out


def peer_store_measure(devices, cfg, steps, warmup):
    """The frame of config `cfg` on `devices` from this one process: rt_mgpu_render, image-addressed peer stores into device 0."""
    from opengl_raytracing_amd import host, scenes
    sc = scenes.make_scene(cfg, host.generate_aabb)
    p = sc.params()
    with host.RayTracer(devices[0]) as rt:
        rt.load(sc)
        rays = rt.count_rays(p)
        rt.render(p)
        ref = rt.readback()
    with host.MultiGpuRayTracer(devices, strip_rows=8) as mg:
        mg.load(sc)
        for _ in range(max(3, warmup)):
            mg.render(p)
        mg.sync()
        got = mg.readback()
        same = all(np.array_equal(g.view(np.uint8), r.view(np.uint8)) for g, r in zip(got, ref))
        t0 = time.perf_counter()
        for _ in range(steps):
            mg.render(p)
        mg.sync()
        el = time.perf_counter() - t0
        lat = 0.0
        n_lat = max(3, min(steps, 10))
        for _ in range(n_lat):
            c0 = time.perf_counter()
            mg.render(p)
            mg.sync()
            lat += time.perf_counter() - c0
        per_dev = [round(float(x), 4) for x in mg.last_ms()]
    return {"workload": workload_name(cfg, sc), "n_devices": len(devices), "devices": list(devices), "steps": steps,
            "ms_per_step": round(el / steps * 1e3, 4), "value_mray_s": round(rays * steps / el / 1e6, 1),
            "frame_latency_ms": round(lat / n_lat * 1e3, 4), "last_frame_kernel_ms_per_device": per_dev,
            "assembled_frame_equals_single_gpu_render": bool(same),
            "how": "one process, one context per device, interleaved 8-row strips, every kernel stores gColor / gPosition / "
                   "gNormal at image addresses of device 0's frame (peer access over xGMI); frames issued back to back, "
                   "wall clock over `steps` frames between two device syncs"}


def peer_store_child(args):
    """`bench.py --peer-store-child N`: prints one JSON object; run by rank 0 of an N > 1 bench (or by hand)."""
    import torch
    devs = ([int(x) for x in args.peer_store_devices.split(",")] if args.peer_store_devices
            else list(range(args.peer_store_child)))
    n_vis = torch.cuda.device_count()
    if max(devs) >= n_vis:
        print(json.dumps({"error": f"{len(devs)} devices asked for, {n_vis} visible to this process"}), flush=True)
        return
    out = {}
    for c in [args.config] + [int(x) for x in args.extra_configs.split(",") if x.strip() and x.strip().lower() != "none"]:
        k = args.steps if c == args.config else max(3, min(args.steps, 10))
        try:
            out[f"c{c}"] = peer_store_measure(devs, c, k, args.warmup if c == args.config else 2)
        except Exception as e:       # reported, never fatal for the bench line
            out[f"c{c}"] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)


def run_peer_store_child(world, cfg, extra, steps, warmup, timeout_s=300, devices=None):
    """Rank 0: the peer-store measurement in a child process (own HIP contexts on all N devices) under a timeout.
    devices: explicit list (the one-GPU rehearsal passes [0] * N: control flow only)."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    cmd = [sys.executable, os.path.abspath(__file__), "--peer-store-child", str(world), "--config", str(cfg),
           "--extra-configs", ",".join(str(c) for c in extra) or "none", "--steps", str(steps), "--warmup", str(warmup)]
    if devices:
        cmd += ["--peer-store-devices", ",".join(str(d) for d in devices)]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"child exit {r.returncode}: {(r.stderr or r.stdout)[-400:]}"}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {"error": f"child exceeded {timeout_s} s and was killed"}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, help="headline BASELINE.json config 1..5 (default 2 = the metric's)")
    ap.add_argument("--extra-configs", default="3,4,5",
                    help="further configs measured after the headline and reported under `configs_extra` "
                         "(N = 1: comma list, default 3,4,5; N > 1: default 5 = the 8K frame the tiling was designed for); 'none' skips")
    ap.add_argument("--no-modes", action="store_true",
                    help="N = 1: skip the free-running-frameCount and raster-order (cold) measurements, so that every launch of the "
                         "dominant kernel in the process belongs to the headline loop (profiles/run_profile.sh uses this: "
                         "rocprofv3's per-kernel average is then the headline's step_ms_device minus the launch gaps)")
    ap.add_argument("--strip-rows", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=1, help="1 = packet kernel (default), 0 = exhaustive loop")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames overlapping on the device, each on its own render stream (1..4; default 3 at N>1, "
                         "1 at N=1 so that the per-kernel duration is the rocprofv3 one)")
    ap.add_argument("--root-weight", type=int, default=0,
                    help="N>1: strips rank 0 owns per cycle (others own 1); 0 = autotune over 1,2,3,4,6 on untimed frames")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 control-flow rehearsal on a 1-GPU box: every rank uses cuda:0 and the gather goes "
                         "through gloo on host copies (RCCL refuses two ranks on one device). Not a measurement.")
    ap.add_argument("--peer-store-child", type=int, default=0,
                    help="internal / by hand: measure the single-process peer-store path (rt_mgpu_*) on devices 0..N-1 and print it")
    ap.add_argument("--peer-store-devices", default="", help="with --peer-store-child: explicit device list, e.g. 0,0,0,0 "
                    "(the N-way plan on one GPU: a control-flow rehearsal, not a measurement)")
    ap.add_argument("--no-peer-store", action="store_true", help="N > 1: skip the peer-store measurement")
    args = ap.parse_args()

    if args.peer_store_child:
        if args.extra_configs == "3,4,5":
            args.extra_configs = "5"
        return peer_store_child(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    src_hash = kernel_source_hash()
    extra = [] if args.extra_configs.strip().lower() in ("none", "") else [int(x) for x in args.extra_configs.split(",") if x.strip()]
    extra = [c for c in extra if c != args.config]

    # ------------------------------------------------------------------------------------------------ N = 1
    if world == 1 and (args.frames_in_flight or 0) <= 1:
        r = measure_single(args.config, args.steps, args.warmup, args.variant, local_rank, with_modes=not args.no_modes)
        rec = single_record(args.config, r, args.steps, args.warmup, src_hash)
        sc = r["sc"]
        out = {"metric": "Mray/s", "value": rec["value_mray_s"], "unit": "Mray/s", "n_gpus": 1, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "value_counts": "rays_reference_per_frame (intersectObjects calls of the reference shader; the CPU baselines' unit)",
               "config": {"workload": rec["workload"], "width": r["W"], "height": r["H"], "max_ray_depth": sc.max_ray_depth,
                          "parallelism": "1 GPU", "frames_in_flight": 1,
                          "rays_per_frame": r["rays_ref"], "rays_per_pixel": round(r["rays_ref"] / (r["W"] * r["H"]), 3)},
               "kernel_source_hash": src_hash}
        for k, v in rec.items():
            if k not in ("workload", "steps", "warmup", "ms_per_step", "value_mray_s"):
                out[k] = v
        # secondary, work-equivalent flop figure; kept for continuity with round 1
        rays_per_shade = 1 + sum((int(l["pcfSamples"]) if int(l["shadowType"]) == 1 else
                                  16 + int(l["pcfSamples"]) if int(l["shadowType"]) == 2 else 0) for l in sc.lights)
        f_alg = r["rays_ref"] * (len(sc.objects) * 25 + 40) + (r["rays_ref"] / rays_per_shade) * len(sc.lights) * 120
        out["valu_flops_equiv"] = {"TFLOPs_equivalent": round(f_alg / (r["step_ms_dev"] * 1e-3) / 1e12, 1),
                                   "peak_TFLOPs": FP32_VALU_PEAK_TFLOPS,
                                   "note": "flops of the EXHAUSTIVE traversal (SURVEY.md 8(d)); packet culling skips most of "
                                           "them, so this is work-equivalent throughput, not a utilisation"}
        if not args.no_cpu_baseline:
            from oracle import binding as O   # checker / reported baseline only
            O.load()
            threads = host_threads()
            base = sc.params()
            t_cpu, n_frames = 0.0, 0
            O.render(sc, sc.params(width=r["W"] // 8, height=r["H"] // 8))   # warm
            while t_cpu < 10.0 and n_frames < 8:
                c0 = time.perf_counter()
                _, _, _, r_cpu = O.render(sc, base, nthreads=threads)
                t_cpu += time.perf_counter() - c0
                n_frames += 1
            out["cpu_baseline"] = {"value": round(r_cpu * n_frames / t_cpu / 1e6, 2), "unit": "Mray/s", "cores": threads,
                                   "kind": "port",
                                   "sample": f"{n_frames} whole frame(s) of the same workload ({r_cpu} rays each) in "
                                             f"{t_cpu:.1f} s; scalar fp32 C restatement (oracle/rt_oracle.c), OpenMP over rows",
                                   "ms_per_frame": round(t_cpu / n_frames * 1e3, 1)}
        try:      # the reference's own GLSL on llvmpipe: static record from the build container (cannot run on the box)
            ref = json.load(open(os.path.join(REPO, "profiles", "reference_llvmpipe_timing.json")))
            out["cpu_baseline_reference"] = ref.get(f"c{args.config}", ref)
        except Exception:
            out["cpu_baseline_reference"] = None
        if extra:
            out["configs_extra"] = {}
            for c in extra:
                k = max(5, min(args.steps, 20))
                rr = measure_single(c, k, max(2, min(args.warmup, 5)), args.variant, local_rank, with_modes=False)
                out["configs_extra"][f"c{c}"] = single_record(c, rr, k, max(2, min(args.warmup, 5)), src_hash)
        print(json.dumps(out), flush=True)
        return

    # ------------------------------------------------------------------------------------------------ N > 1 (or F > 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from opengl_raytracing_amd import dist as D
    from opengl_raytracing_amd import host, scenes

    dev = torch.device("cuda", local_rank)
    red_dev = "cpu" if args.rehearse_on_one_gpu else dev
    # Dedicated (non-null) torch streams.  The render kernel is launched on a render stream through the ABI; at
    # N > 1 the RCCL gather and the pack / re-assembly kernels are issued from `s_comm`:
    #   s_render[k%F]:  [wait slot k-F free] render k -> E_render[k%F]
    #   s_comm       :  wait E_render[k%F]; peers: pack k (40 -> 30 B/px); gather k; rank 0: unpack + de-interleave k
    # Consecutive frames rotate over F render streams (F frames in flight): a rank's share of a 1080p frame is a
    # single round of resident waves whose length is its slowest tile, so frame k+1 fills the CUs that frame k's
    # short tiles have already left.  The timed region brackets K complete frames (render + gather + re-assembly).
    F = max(1, min(4, args.frames_in_flight or (1 if world == 1 else 3)))
    s_renders = [torch.cuda.Stream(device=dev) for _ in range(F)]
    s_render = s_renders[0]
    s_comm = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(s_render)
    assert all(st.cuda_stream != 0 for st in s_renders) and s_comm.cuda_stream != 0
    stream = s_render

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_config(cfg, steps, warmup, autotune):
        sc = scenes.make_scene(cfg, host.generate_aabb)
        W, H = sc.width, sc.height
        base = sc.params()
        rt = host.RayTracer(local_rank)
        rt.load(sc)
        rt.set_variant(args.variant)
        strip_rows = args.strip_rows or D.default_strip_rows(H, world, len(sc.objects))
        full = None
        if world > 1 and rank == 0:
            full = [torch.empty((H, W, 4), dtype=dt, device=dev) for dt in (torch.float32, torch.float32, torch.float16)]

        class Pipeline:
            """One strip plan's buffers + the per-frame step.  root_weight = strips rank 0 owns per cycle: its rows
            stay on rank 0 (never packed, never sent), so a heavier root trades its own render time against the
            bytes converging on its inbound xGMI links."""

            def __init__(self, root_weight):
                self.plan = D.StripPlan(W, H, strip_rows, world, root_weight) if world > 1 else D.StripPlan(W, H, H, 1)
                plan = self.plan
                self.p = plan.params(base, rank) if world > 1 else base
                me = rank if world > 1 else None
                self.bufs = [D.alloc_rank_buffer(plan, dev, me) for _ in range(F)]
                self.views = [D.surface_views(b, plan, me) for b in self.bufs]
                # peers: real wire buffers; rank 0: one placeholder (torch's gather wants a contribution from the root)
                self.wires = [D.alloc_wire_buffer(plan, dev) for _ in range(F if rank > 0 else 1)] if world > 1 else None
                self.gathered = ([torch.empty((world, plan.wire_bytes), dtype=torch.uint8, device=dev) for _ in range(2)]
                                 if world > 1 and rank == 0 else [None, None])
                self.ev_render = [torch.cuda.Event() for _ in range(F)]
                self.ev_free = [torch.cuda.Event() for _ in range(F)]     # slot's surfaces may be rendered into again
                self.k = 0

            def step(self):
                k = self.k
                self.k += 1
                plan, p = self.plan, self.p
                if world == 1:          # F > 1 (opt-in): consecutive frames on alternating streams and surfaces
                    c, q, n = self.views[k % F]
                    rt.render_to(p, c.data_ptr(), q.data_ptr(), n.data_ptr(), stream=s_renders[k % F].cuda_stream)
                    return
                b, g2 = k % F, k & 1      # g2: rank 0's gather target (s_comm is in order, two are plenty)
                c, q, n = self.views[b]
                sr = s_renders[b]
                if k >= F:
                    sr.wait_event(self.ev_free[b])
                rt.render_to(p, c.data_ptr(), q.data_ptr(), n.data_ptr(), stream=sr.cuda_stream)
                self.ev_render[b].record(sr)
                with torch.cuda.stream(s_comm):
                    s_comm.wait_event(self.ev_render[b])
                    if rank > 0:          # surfaces -> 30 B/pixel wire buffer; once packed the slot is free again
                        wire = D.pack_wire_hip(rt, self.views[b], self.wires[b], plan, stream=s_comm.cuda_stream)
                        self.ev_free[b].record(s_comm)
                    else:
                        wire = self.wires[0]
                    if args.rehearse_on_one_gpu:
                        s_comm.synchronize()
                        g = D.gather_wire(wire.cpu(), plan, rank)
                        if rank == 0:
                            self.gathered[g2].copy_(g)
                            g = self.gathered[g2]
                    else:
                        g = D.gather_wire(wire, plan, rank, out=self.gathered[g2])           # ONE RCCL gather per frame
                    if rank == 0:         # peers' strips from the wire, the root's own rows from its local surfaces
                        D.unpack_wire_hip(rt, g, plan, outs=full, root_views=self.views[b], stream=s_comm.cuda_stream)
                        self.ev_free[b].record(s_comm)

            def run(self, n_frames):
                """n_frames complete frames, drained; wall seconds (max over ranks)."""
                fence()
                t0 = time.perf_counter()
                for _ in range(n_frames):
                    self.step()
                fence()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
                if world > 1:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item())

        # Root share: autotuned on untimed frames unless --root-weight fixes it.  Every rank sees the same
        # max-over-ranks times, so every rank picks the same plan.
        tune = None
        root_weight = 1
        if world > 1:
            if args.root_weight > 0:
                root_weight = args.root_weight
            elif autotune:
                tune = {}
                n_t = 36 if W * H <= 1920 * 1080 else 6
                for w0 in (1, 2, 3, 4, 6):
                    trial = Pipeline(w0)
                    trial.run(max(2, n_t // 3))                       # LPT order, clocks, RCCL channels
                    tune[w0] = round(min(trial.run(n_t), trial.run(n_t)) / n_t * 1e3, 4)   # ms / frame
                    del trial
                root_weight = min(tune, key=tune.get)
        pipe = Pipeline(root_weight)
        plan, p = pipe.plan, pipe.p
        s_col, s_pos, s_nrm = pipe.views[0]

        # exact ray counts of this rank's pixels (instrumented launches, outside the timed region)
        counts = torch.tensor([rt.count_rays(p), rt.count_rays_traced(p)], dtype=torch.int64, device=red_dev)
        my_rays = int(counts[0].item())
        if world > 1:
            dist.all_reduce(counts)
        frame_rays, frame_traced = int(counts[0].item()), int(counts[1].item())

        for _ in range(warmup):
            pipe.step()
        fence()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for st in s_renders[1:]:
            st.wait_event(ev0)
        for _ in range(steps):
            pipe.step()
        for st in s_renders[1:]:
            s_render.wait_stream(st)
        if world > 1:
            s_render.wait_stream(s_comm)      # the last frame's gather + re-assembly belongs to the timed region
        ev1.record(stream)
        fence()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

        # this rank's render share: kernel launches only, back to back on one stream (HIP events)
        kev0, kev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        kev0.record(stream)
        for _ in range(steps):
            rt.render_to(p, s_col.data_ptr(), s_pos.data_ptr(), s_nrm.data_ptr(), stream=stream.cuda_stream)
        kev1.record(stream)
        torch.cuda.synchronize()
        kernel_ms = kev0.elapsed_time(kev1) / steps
        step_ms_dev = ev0.elapsed_time(ev1) / steps

        # frame LATENCY (one frame at a time, nothing overlapped) and its parts, per rank
        lat = None
        if world > 1:
            n_lat = max(3, min(steps, 10))
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            t_lat, t_render, t_comm = 0.0, 0.0, 0.0
            for _ in range(n_lat):
                fence()
                c0 = time.perf_counter()
                e[0].record(s_render)
                for st in s_renders[1:]:
                    st.wait_event(e[0])
                pipe.step()
                for st in s_renders[1:]:
                    s_render.wait_stream(st)
                e[1].record(s_render)             # this frame's render share is done
                s_render.wait_stream(s_comm)
                e[2].record(s_render)             # ... and its pack / gather / re-assembly
                fence()
                t_lat += time.perf_counter() - c0
                t_render += e[0].elapsed_time(e[1])
                t_comm += e[1].elapsed_time(e[2])
            per = torch.tensor([t_render / n_lat, t_comm / n_lat], dtype=torch.float64, device=red_dev)
            allper = [torch.zeros_like(per) for _ in range(world)]
            dist.all_gather(allper, per)
            tl = torch.tensor([t_lat / n_lat], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tl, op=dist.ReduceOp.MAX)
            lat = {"frame_latency_ms": round(float(tl.item()) * 1e3, 4),
                   "render_ms_per_rank": [round(float(x[0]), 4) for x in allper],
                   "pack_gather_unpack_ms_per_rank": [round(float(x[1]), 4) for x in allper],
                   "note": "one frame at a time between barriers (no frames in flight): render share, then pack + RCCL "
                           "gather + re-assembly as seen by each rank's streams"}

        # untimed self-check at N > 1: the assembled frame must equal a single-GPU render bit for bit
        assembled_ok = None
        if world > 1 and rank == 0:
            one = D.StripPlan(W, H, H, 1)
            ref_buf = D.alloc_rank_buffer(one, dev)
            rc, rp, rn = D.surface_views(ref_buf, one)
            rt.render_to(base, rc.data_ptr(), rp.data_ptr(), rn.data_ptr(), stream=stream.cuda_stream)
            torch.cuda.synchronize()
            assembled_ok = bool(torch.equal(full[0].view(torch.int32), rc.view(torch.int32)) and
                                torch.equal(full[1].view(torch.int32), rp.view(torch.int32)) and
                                torch.equal(full[2].view(torch.int16), rn.view(torch.int16)))
        rec = None
        if rank == 0:
            n_px = W * H
            my_px = plan.local_rows(rank) * W if world > 1 else n_px
            rec = {"workload": workload_name(cfg, sc), "steps": steps, "warmup": warmup,
                   "ms_per_step": round(elapsed / steps * 1e3, 4),
                   "value_mray_s": round(frame_rays * steps / elapsed / 1e6, 1),
                   "mray_s_traced": round(frame_traced * steps / elapsed / 1e6, 1),
                   "mpx_per_s": round(n_px * steps / elapsed / 1e6, 1),
                   "rays_reference_per_frame": frame_rays, "rays_traced_per_frame": frame_traced,
                   "rank0_render_share_kernel_ms": round(kernel_ms, 4), "step_ms_device": round(step_ms_dev, 4),
                   "width": W, "height": H, "max_ray_depth": sc.max_ray_depth,
                   "parallelism": (f"{world} ranks x interleaved {strip_rows}-row strips, rank 0 owns {root_weight} of every "
                                   f"{root_weight + world - 1} (its rows stay local), one RCCL gather/frame of the 30 B/px "
                                   f"wire format, {F} frames in flight (render streams), gather k overlapped with the "
                                   f"renders of the following frames") if world > 1 else f"1 GPU, {F} frames in flight",
                   "frames_in_flight": F, "root_weight": root_weight if world > 1 else None,
                   "root_weight_autotune_ms_per_frame": tune,
                   "bytes_into_rank0_per_frame": (plan.wire_bytes * (world - 1)) if world > 1 else 0,
                   "rank0_pixels": my_px, "rank0_rays_reference": my_rays,
                   "assembled_frame_equals_single_gpu_render": assembled_ok, "latency": lat}
        rt.close()
        return rec

    head = run_config(args.config, args.steps, args.warmup, autotune=True)
    extras = {}
    if world > 1 and args.extra_configs == "3,4,5":
        extra = [5] if args.config != 5 else []          # the frame the tiling was designed for
    for c in extra:
        k = max(3, min(args.steps, 10))
        extras[f"c{c}"] = run_config(c, k, 2, autotune=False)
    peer = None
    if world > 1 and not args.no_peer_store:
        # the other ranks wait on the rendezvous store (host side): an RCCL barrier would spin on their GPUs meanwhile
        torch.cuda.synchronize()
        dist.barrier()
        store = None
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:
            pass
        if rank == 0:
            peer = run_peer_store_child(world, args.config, extra, args.steps, args.warmup,
                                        devices=[0] * world if args.rehearse_on_one_gpu else None)
            if store is not None:
                store.set("bench_peer_store_done", "1")
        elif store is not None:
            import datetime
            try:
                store.wait(["bench_peer_store_done"], datetime.timedelta(seconds=420))
            except Exception:
                pass
        if store is None:
            dist.barrier()
    if rank == 0:
        out = {"metric": "Mray/s", "value": head["value_mray_s"], "unit": "Mray/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "value_counts": "rays_reference_per_frame (intersectObjects calls of the reference shader; the CPU baselines' unit)",
               "config": {"workload": head["workload"], "width": head["width"], "height": head["height"],
                          "max_ray_depth": head["max_ray_depth"], "parallelism": head["parallelism"],
                          "frames_in_flight": head["frames_in_flight"], "root_weight": head["root_weight"],
                          "root_weight_autotune_ms_per_frame": head["root_weight_autotune_ms_per_frame"],
                          "rays_per_frame": head["rays_reference_per_frame"],
                          "rays_per_pixel": round(head["rays_reference_per_frame"] / (head["width"] * head["height"]), 3)},
               "rehearsal": bool(args.rehearse_on_one_gpu), "kernel_source_hash": src_hash,
               "roofline": {"bound": "valu-issue", "achieved": None, "peak": round(VALU_ISSUE_PEAK, 1),
                            "unit": "G wave64 VALU instr/s", "frac": None, "traffic": None,
                            "note": "per-kernel roofline is reported by the N = 1 run (one rank renders a strip subset here)"}}
        for k, v in head.items():
            if k not in ("workload", "steps", "warmup", "ms_per_step", "value_mray_s", "width", "height", "max_ray_depth",
                         "parallelism", "frames_in_flight", "root_weight", "root_weight_autotune_ms_per_frame"):
                out[k] = v
        if extras:
            out["configs_extra"] = extras
        if peer is not None:
            out["peer_store"] = peer
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
