#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its named workload, on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one frame: one pass of the ray-tracing hot path over the workload's pixels, with
the scene, textures and output surfaces already resident in HBM (the reference re-uploads its
<= 4 KB of SSBOs per frame; the PCIe-inclusive figure is in DESIGN.md, never `value`).

Workload (N = 1 and N > 1 alike): BASELINE.json configs[1] -- 1920x1080, 16 spheres + 2
planes, 3 lights (point / directional / area), MAX_RAY_DEPTH 4, PCF x4 shadows, synthetic
scene of SURVEY.md 8(d).  N > 1 splits that one frame into interleaved 16-row strips, one
process per GPU, and assembles the image on rank 0 with one RCCL gather per surface + a copy
kernel -- all inside the timed step ("scaling": "strong").

Rank 0 prints ONE JSON line: metric Mray/s (rays = intersectObjects calls, counted exactly by
an instrumented launch before the timed region), ms_per_step, `roofline` (algorithmic bytes
of SURVEY.md 8(d) / live HIP-event kernel time) and, at N = 1, `cpu_baseline` (the oracle --
the scalar CPU restatement -- timed on this box's host cores on whole frames of the same
workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s HBM3E peak
FP32_VALU_PEAK_TFLOPS = 157.3


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config index 1..5 (default 2 = the metric's)")
    ap.add_argument("--strip-rows", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=1, help="1 = packet kernel (default), 0 = exhaustive loop")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames overlapping on the device, each on its own render stream (1..4; default 3 at N>1, "
                         "1 at N=1 so that the per-kernel duration is the rocprofv3 one; N=1 with 2-3 measures 0.47 ms/frame)")
    ap.add_argument("--root-weight", type=int, default=0,
                    help="N>1: strips rank 0 owns per cycle (others own 1); 0 = autotune over 1,2,3,4,6 on untimed frames")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 control-flow rehearsal on a 1-GPU box: every rank uses cuda:0 and the gather goes "
                         "through gloo on host copies (RCCL refuses two ranks on one device). Not a measurement.")
    args = ap.parse_args()

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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from opengl_raytracing_amd import dist as D
    from opengl_raytracing_amd import host, scenes

    sc = scenes.make_scene(args.config, host.generate_aabb)
    W, H = sc.width, sc.height
    base = sc.params()
    rt = host.RayTracer(local_rank)
    rt.load(sc)
    rt.set_variant(args.variant)

    strip_rows = args.strip_rows or D.default_strip_rows(H, world, len(sc.objects))
    dev = torch.device("cuda", local_rank)
    red_dev = "cpu" if args.rehearse_on_one_gpu else dev
    # Dedicated (non-null) torch streams.  The render kernel is launched on a render stream through the ABI; at
    # N > 1 the RCCL gather and the pack / re-assembly kernels are issued from `s_comm`:
    #   s_render[k%F]:  [wait slot k-F free] render k -> E_render[k%F]
    #   s_comm       :  wait E_render[k%F]; peers: pack k (40 -> 30 B/px); gather k; rank 0: unpack + de-interleave k
    # At N > 1 consecutive frames rotate over F render streams (F frames in flight): a rank's share of a 1080p
    # frame is a single round of resident waves whose length is its slowest tile (0.25 ms at N = 4 and 8 alike,
    # tools/gpu_strip_scaling.py), so frame k+1 fills the CUs that frame k's short tiles have already left
    # (N = 8 share on one GPU: 0.25 ms/frame with F = 1, 0.13 with F = 2, 0.09 with F = 3).
    # The timed region still brackets K complete frames (render + gather + re-assembly, drained).
    F = max(1, min(4, args.frames_in_flight or (1 if world == 1 else 3)))
    s_renders = [torch.cuda.Stream(device=dev) for _ in range(F)]
    s_render = s_renders[0]
    s_comm = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(s_render)
    assert all(st.cuda_stream != 0 for st in s_renders) and s_comm.cuda_stream != 0
    stream = s_render
    full = None
    if world > 1 and rank == 0:
        full = [torch.empty((H, W, 4), dtype=dt, device=dev) for dt in (torch.float32, torch.float32, torch.float16)]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
        else:
            tune = {}
            for w0 in (1, 2, 3, 4, 6):
                trial = Pipeline(w0)
                trial.run(12)                                   # LPT order, clocks, RCCL channels
                tune[w0] = round(min(trial.run(36), trial.run(36)) / 36 * 1e3, 4)   # ms / frame
                del trial
            root_weight = min(tune, key=tune.get)
    pipe = Pipeline(root_weight)
    plan, p = pipe.plan, pipe.p
    s_col, s_pos, s_nrm = pipe.views[0]
    step = pipe.step

    # exact ray count of this rank's pixels (instrumented launch, outside the timed region)
    my_rays = rt.count_rays(p)
    rays_t = torch.tensor([my_rays], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(rays_t)
    frame_rays = int(rays_t.item())

    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for st in s_renders[1:]:
        st.wait_event(ev0)
    for _ in range(args.steps):
        step()
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

    # dominant kernel's average launch duration: HIP events on the launch stream, kernel only
    # (a second pass of K back-to-back launches so that at N>1 the gather is not inside it)
    kev0, kev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    kev0.record(stream)
    for _ in range(args.steps):
        rt.render_to(p, s_col.data_ptr(), s_pos.data_ptr(), s_nrm.data_ptr(), stream=stream.cuda_stream)
    kev1.record(stream)
    torch.cuda.synchronize()
    kernel_ms = kev0.elapsed_time(kev1) / args.steps
    step_ms_dev = ev0.elapsed_time(ev1) / args.steps

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

    if rank == 0:
        n_px = W * H
        ms_per_step = elapsed / args.steps * 1e3
        value = frame_rays * args.steps / elapsed / 1e6
        my_px = plan.local_rows(rank) * W if world > 1 else n_px
        b_alg = algorithmic_bytes(my_rays, len(sc.objects), len(sc.lights), my_px, sc.noise is not None, 0)
        achieved = b_alg / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tpath) and world == 1:
            try:
                traffic = json.load(open(tpath)).get(f"c{args.config}", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        compulsory = (my_px * 40 + len(sc.objects) * 176 + len(sc.lights) * 96) / (kernel_ms * 1e-3) / 1e9
        # secondary, honest limiter: algorithmic flops of the exhaustive traversal
        #   F_alg = R*(nObj*25 + 40) + S*nLt*120,  S = shading points ~ R / (1 + sum_l s_l)
        rays_per_shade = 1 + sum((int(l["pcfSamples"]) if int(l["shadowType"]) == 1 else
                                  16 + int(l["pcfSamples"]) if int(l["shadowType"]) == 2 else 0) for l in sc.lights)
        f_alg = my_rays * (len(sc.objects) * 25 + 40) + (my_rays / rays_per_shade) * len(sc.lights) * 120
        valu_tflops = f_alg / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "Mray/s", "value": round(value, 1), "unit": "Mray/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C{args.config}: {W}x{H}, {len(sc.objects)} objects "
                                   f"({int((sc.objects['type'] == 0).sum())} spheres + {int((sc.objects['type'] == 1).sum())} planes), "
                                   f"{len(sc.lights)} lights, depth {sc.max_ray_depth}, "
                                   f"{'PCSS' if int(sc.lights['shadowType'][0]) == 2 else 'PCF x4'} shadows",
                       "width": W, "height": H, "max_ray_depth": sc.max_ray_depth,
                       "parallelism": (f"{world} ranks x interleaved {strip_rows}-row strips, rank 0 owns {root_weight} of every "
                                       f"{root_weight + world - 1} (its rows stay local), one RCCL gather/frame of the 30 B/px "
                                       f"wire format ({plan.wire_bytes * (world - 1)} B into rank 0), {F} frames in flight "
                                       f"(render streams), gather k overlapped with the renders of the following frames")
                                      if world > 1 else "1 GPU",
                       "frames_in_flight": F, "root_weight": root_weight if world > 1 else None, "root_weight_autotune_ms_per_frame": tune,
                       "rays_per_frame": frame_rays, "rays_per_pixel": round(frame_rays / n_px, 3)},
            "mpx_per_s": round(n_px * args.steps / elapsed / 1e6, 1), "rehearsal": bool(args.rehearse_on_one_gpu),
            "assembled_frame_equals_single_gpu_render": assembled_ok,
            "kernel_ms": round(kernel_ms, 4), "step_ms_device": round(step_ms_dev, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 3), "traffic": traffic,
                         "algorithmic_bytes_per_launch": b_alg,
                         "note": "algorithmic bytes = rays*nObj*176 + px*40 + scene (SURVEY.md 8(d)); the stream is "
                                 "served from LDS, so achieved may exceed the HBM peak -- it is not physical bandwidth",
                         "compulsory_only_GBps": round(compulsory, 1)},
            "roofline_valu": {"bound": "fp32 valu", "achieved": round(valu_tflops, 1), "peak": FP32_VALU_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": round(valu_tflops / FP32_VALU_PEAK_TFLOPS, 3),
                              "note": "algorithmic flops of the exhaustive traversal (SURVEY.md 8(d)); packet culling "
                                      "skips most of them, so this is work-equivalent throughput"},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import binding as O   # checker / reported baseline only
            O.load()
            threads = host_threads()
            t_cpu, n_frames = 0.0, 0
            O.render(sc, sc.params(width=W // 8, height=H // 8))   # warm
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
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    rt.close()


if __name__ == "__main__":
    main()
