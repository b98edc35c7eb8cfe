"""N > 1 path on CPU: world_size-2 and -3 `gloo` process groups run the same strip plan,
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


def _worker(rank, world, port, strip_rows, w, h, out_path):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from opengl_raytracing_amd import dist as D
    from opengl_raytracing_amd import host, scenes
    from oracle import binding as O

    sc = scenes.make_scene(2, host.generate_aabb)
    base = sc.params(width=w, height=h)
    plan = D.StripPlan(w, h, strip_rows, world)
    p = plan.params(base, rank)
    col, pos, nrm, rays = O.render(sc, p, nthreads=2)
    assert col.shape[0] == plan.max_local_rows
    buf = D.alloc_rank_buffer(plan, "cpu")
    vc, vp, vn = D.surface_views(buf, plan)
    vc.copy_(torch.from_numpy(col))
    vp.copy_(torch.from_numpy(pos))
    vn.view(torch.int16).copy_(torch.from_numpy(nrm.view(np.int16)))
    g = D.gather_rank_buffers(buf, plan, rank)
    total = torch.tensor([rays], dtype=torch.int64)
    dist.all_reduce(total)
    if rank == 0:
        full = D.deinterleave_torch(g, plan)
        np.savez(out_path, color=full[0].numpy(), pos=full[1].numpy(), normal=full[2].view(torch.int16).numpy(),
                 rays=int(total.item()))
    else:
        assert g is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,strip_rows,size", [(2, 16, (96, 70)), (3, 8, (64, 45))])
def test_gloo_strip_gather_reproduces_single_frame(tmp_path, world, strip_rows, size):
    from opengl_raytracing_amd import host, scenes
    from oracle import binding as O
    w, h = size
    out = str(tmp_path / "full.npz")
    mp.spawn(_worker, args=(world, _free_port(), strip_rows, w, h, out), nprocs=world, join=True)
    got = np.load(out)
    sc = scenes.make_scene(2, host.generate_aabb)
    col, pos, nrm, rays = O.render(sc, sc.params(width=w, height=h))
    assert np.array_equal(got["color"], col, equal_nan=True)
    assert np.array_equal(got["pos"], pos, equal_nan=True)
    assert np.array_equal(got["normal"], nrm.view(np.int16))
    assert int(got["rays"]) == rays
