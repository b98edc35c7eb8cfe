import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference + Mesa llvmpipe (build container only)")


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    """-m gpu tests must never pass on a silent fallback: without a GPU they are skipped
    (the driver deselects them here anyway), with a GPU they load the in-tree HIP library."""
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# ---- comparison helpers shared by the parity tests -------------------------------------------
def compare_surface(a, b, rtol=1e-4, atol=1e-6):
    """Per-pixel parity metric of SURVEY.md 8(c): |a-b| <= rtol*max(|a|,|b|) + atol per channel,
    NaN == NaN.  Returns dict(pass_frac, exact_frac, n_fail, ok_mask)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    both_nan = np.isnan(a) & np.isnan(b)
    with np.errstate(invalid="ignore"):
        err = np.abs(a.astype(np.float64) - b.astype(np.float64))
        tol = rtol * np.maximum(np.abs(a), np.abs(b)).astype(np.float64) + atol
        same = (a == b) | both_nan | (np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b)))
        ok = (err <= tol) | same
    okpx = ok.all(axis=-1)
    expx = same.all(axis=-1)
    return dict(pass_frac=float(okpx.mean()), exact_frac=float(expx.mean()), n_fail=int((~okpx).sum()),
                ok_mask=okpx, exact_mask=expx)


def bits_equal(a, b):
    """Bitwise equality with NaN payloads ignored."""
    a = np.asarray(a)
    b = np.asarray(b)
    if a.dtype.kind == "f":
        return bool((((a == b) | (np.isnan(a) & np.isnan(b)))).all())
    return bool((a == b).all())


class GoldenScene:
    """A scenes.Scene-like object rebuilt from the bytes stored in a golden fixture."""

    def __init__(self, npz):
        from opengl_raytracing_amd import layout as L
        from opengl_raytracing_amd import scenes
        self.objects = np.frombuffer(npz["objects"].tobytes(), dtype=L.OBJECT_DTYPE).copy()
        self.lights = np.frombuffer(npz["lights"].tobytes(), dtype=L.LIGHT_DTYPE).copy()
        self.frame_count = int(npz["frame_count"])
        self.noise = scenes.hash_noise() if int(npz["has_noise"]) else None
        self.use_skybox = bool(int(npz["has_skybox"]))
        self.skybox = scenes.procedural_skybox(512) if self.use_skybox else None


def load_golden(name):
    path = os.path.join(GOLDEN_DIR, f"{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"golden fixture {name}.npz not generated")
    return np.load(path)


def params_from_bytes(b):
    from opengl_raytracing_amd import layout as L
    return L.RtParams.from_buffer_copy(bytes(np.asarray(b, dtype=np.uint8)))


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.load()
    return binding


@pytest.fixture(scope="session")
def host():
    from opengl_raytracing_amd import host as h
    h.load_library()
    return h


@pytest.fixture(scope="session", params=[1, 0], ids=["packet", "exhaustive"])
def tracer(host, request):
    """One device context per kernel variant for the whole GPU session (fails loudly if the HIP
    path is unusable).  Every parity test runs against both the default wavefront-packet kernel
    (variant 1) and the exhaustive per-lane loop (variant 0)."""
    rt = host.RayTracer(0)
    rt.set_variant(request.param)
    rt.variant = request.param
    yield rt
    rt.close()
