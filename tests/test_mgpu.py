"""rt_mgpu_*: one frame on N devices from ONE process (include/rt_mi355.h, csrc/rt_mgpu.cpp) -- every device renders its
interleaved strips and its kernel stores them straight into device 0's full-frame surfaces.  On the one-GPU test box the N
"devices" are all device 0: the N-way plan (strip mapping, image-addressed stores, per-device contexts and streams, the event
ordering between them) executes for N up to 8, which the one-process-per-GPU rehearsal could not (the pool allows 6 processes
on a card).  The assembled frame must equal the single-device render bit for bit: pixels are independent."""
import numpy as np
import pytest

from conftest import bits_equal
from opengl_raytracing_amd import layout as L
from opengl_raytracing_amd import scenes

pytestmark = pytest.mark.gpu


def _single(host, sc, p):
    with host.RayTracer(0) as rt:
        rt.load(sc)
        rt.render(p)
        return rt.readback()


@pytest.mark.parametrize("cfg,size", [(2, (642, 371)), (5, (480, 270)), (3, (384, 216))])
def test_n_way_frame_equals_single_device_render(host, cfg, size):
    sc = scenes.make_scene(cfg, host.generate_aabb)
    p = sc.params(width=size[0], height=size[1])
    ref = _single(host, sc, p)
    for n, strip in [(2, 8), (3, 16), (4, 8), (8, 8), (8, 24), (5, 1)]:
        with host.MultiGpuRayTracer([0] * n, strip_rows=strip) as mg:
            mg.load(sc)
            for k in range(3):                 # several frames: the per-device tile schedulers move from raster to measured order
                mg.render(p)
            got = mg.readback()
            for g, r, name in zip(got, ref, ("gColor", "gPosition", "gNormal")):
                assert bits_equal(g, r), f"C{cfg} {size}: {name} of the {n}-way frame (strips of {strip} rows) differs from the single render"
            ms = mg.last_ms()
            assert len(ms) == n and all(m >= 0 for m in ms)


def test_scene_updates_and_consumers_between_n_way_frames(host, oracle):
    """Per-frame rt_mgpu_set_scene (the reference re-uploads its SSBOs every frame) and frames issued back to back without a sync:
    every frame shows the scene that was current when it was issued; a consumer on the root stream sees whole frames."""
    import ctypes
    a = scenes.make_scene(2, host.generate_aabb)
    b = scenes.make_scene(2, host.generate_aabb)
    b.objects["position"][:, 1] += 0.4
    host.generate_aabb(b.objects)
    p = a.params(width=320, height=200)
    want = [oracle.render(a, p), oracle.render(b, p)]
    with host.MultiGpuRayTracer([0] * 4, strip_rows=8) as mg:
        for k in range(6):
            sc = (a, b)[k & 1]
            mg.load(sc)
            mg.render(p)
            col, pos, nrm = mg.readback()
            oc, op, on = want[k & 1][:3]
            assert bits_equal(col, oc) and bits_equal(pos, op) and bits_equal(nrm, on), f"frame {k}"
        # whole-frame parameters only
        bad = L.copy_params(p, regionH=100)
        with pytest.raises(host.RtError):
            mg.render(bad)
