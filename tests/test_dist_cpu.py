"""N > 1 path on CPU: world_size-2, -3, -4 and -8 `gloo` process groups run the same strip plan,
gather and re-assembly code the GPU bench uses (opengl_raytracing_amd/dist.py).  The strips
are rendered by the oracle here (tests may use it; there is no GPU in this container) -- what
is under test is the partition + gather + de-interleave logic, which must reproduce the
single-process frame bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, strip_rows, w, h, out_path, root_weight, cfg=2):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from opengl_raytracing_amd import dist as D
    from opengl_raytracing_amd import host, scenes
    from oracle import binding as O

    sc = scenes.make_scene(cfg, host.generate_aabb)
    base = sc.params(width=w, height=h)
    plan = D.StripPlan(w, h, strip_rows, world, root_weight)
    p = plan.params(base, rank)
    col, pos, nrm, rays = O.render(sc, p, nthreads=2)
    assert col.shape[0] == plan.buffer_rows(rank)
    buf = D.alloc_rank_buffer(plan, "cpu", rank)
    vc, vp, vn = D.surface_views(buf, plan, rank)
    vc.copy_(torch.from_numpy(col))
    vp.copy_(torch.from_numpy(pos))
    vn.view(torch.int16).copy_(torch.from_numpy(nrm.view(np.int16)))
    # the frame through the 30 B/pixel wire format (what bench.py sends over xGMI): peers pack, the root's
    # rows stay local and its gather contribution is a placeholder
    wire = D.pack_wire_torch((vc, vp, vn), plan) if rank > 0 else torch.zeros(plan.wire_bytes, dtype=torch.uint8)
    gw = D.gather_wire(wire, plan, rank)
    total = torch.tensor([rays], dtype=torch.int64)
    dist.all_reduce(total)
    g = None
    if root_weight == 1:                  # surface-format path (rt_deinterleave's layout): equal strips only
        g = D.gather_rank_buffers(buf, plan, rank)
    if rank == 0:
        assert gw.shape == (world, plan.wire_bytes) and plan.wire_bytes < plan.rank_bytes
        wfull = D.unpack_wire_torch(gw, plan, root_views=(vc, vp, vn))
        if root_weight == 1:
            full = D.deinterleave_torch(g, plan)
            for a, b in zip(full, wfull):
                assert torch.equal(a.view(torch.int16), b.view(torch.int16)), "wire path differs from the surface path"
            # and with the root's rows travelling through wire slot 0 like everybody else's
            gw[0].copy_(D.pack_wire_torch((vc, vp, vn), plan))
            for a, b in zip(full, D.unpack_wire_torch(gw, plan)):
                assert torch.equal(a.view(torch.int16), b.view(torch.int16))
        np.savez(out_path, color=wfull[0].numpy(), pos=wfull[1].numpy(), normal=wfull[2].view(torch.int16).numpy(),
                 rays=int(total.item()))
    else:
        assert g is None and gw is None
    dist.barrier()
    dist.destroy_process_group()


# (world, strip rows, size, root weight, config): worlds 2 / 3 on the benchmark scene, and worlds 4 / 8 with the plan shape
# bench.py uses for the 8K frame the tiling was designed for -- C5's scene (256 objects, depth 8, skybox), 8-row strips,
# equal and weighted roots, image heights that leave the last cycle ragged (some ranks own no rows of it)
@pytest.mark.parametrize("world,strip_rows,size,root_weight,cfg", [(2, 16, (96, 70), 1, 2), (3, 8, (64, 45), 1, 2), (2, 8, (64, 70), 2, 2),
                                                                   (3, 8, (48, 61), 3, 2), (4, 8, (40, 75), 2, 5), (8, 8, (32, 100), 1, 5),
                                                                   (8, 8, (24, 150), 3, 5)])
def test_gloo_strip_gather_reproduces_single_frame(tmp_path, world, strip_rows, size, root_weight, cfg):
    from opengl_raytracing_amd import host, scenes
    from oracle import binding as O
    w, h = size
    out = str(tmp_path / "full.npz")
    mp.spawn(_worker, args=(world, _free_port(), strip_rows, w, h, out, root_weight, cfg), nprocs=world, join=True)
    got = np.load(out)
    sc = scenes.make_scene(cfg, host.generate_aabb)
    col, pos, nrm, rays = O.render(sc, sc.params(width=w, height=h))
    assert np.array_equal(got["color"], col, equal_nan=True)
    assert np.array_equal(got["pos"], pos, equal_nan=True)
    assert np.array_equal(got["normal"], nrm.view(np.int16))
    assert int(got["rays"]) == rays
