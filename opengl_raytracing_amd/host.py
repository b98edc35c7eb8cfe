"""ctypes binding of librt_mi355.so -- the Python-side mirror of the reference's dispatch
site (/root/reference/src/ForwardShadingPipeline.cpp:155-182).

``RayTracer`` plays the part of ``ForwardShadingPipline``'s ray-tracing members:

=====================================  ====================================================
reference (C++ / GL)                   here
=====================================  ====================================================
``ssbo.update(); lightSSBO.update()``  ``RayTracer.set_scene(objects, lights)``
``raytracingShader.setXxx(...)``       fields of ``RtParams`` (layout.make_params)
``glBindTexture(CUBE_MAP, ...)``       ``RayTracer.set_skybox(faces)``
blue-noise texture                     ``RayTracer.set_noise(r8)``
``glDispatchCompute + glMemoryBarrier````RayTracer.render(params)``
``glGetTexImage``                      ``RayTracer.readback()``
``gProfiler`` RayTracing stage         ``RayTracer.last_kernel_ms()``
=====================================  ====================================================

There is no CPU fallback: if the HIP library is missing or no GPU is present the calls
raise ``RtError``.
"""
import ctypes
import os

import numpy as np

from . import build as _build
from . import layout as L

_LIB = None

EXPORTS = [
    "rt_create", "rt_destroy", "rt_set_scene", "rt_set_noise", "rt_set_skybox", "rt_render",
    "rt_render_to", "rt_sync", "rt_readback", "rt_get_surfaces", "rt_last_kernel_ms",
    "rt_count_rays", "rt_count_rays_traced", "rt_debug_stats", "rt_debug_stats_ex", "rt_debug_tile_costs", "rt_set_variant", "rt_last_error", "rt_generate_aabb", "rt_camera_vectors",
    "rt_scene_parse", "rt_scene_write", "rt_taa_resolve", "rt_taa_jitter", "rt_bloom", "rt_ssao", "rt_ssao_blur",
    "rt_camera_matrices", "rt_equirect_to_cubemap", "rt_frame", "rt_frame_surfaces", "rt_strip_local_rows", "rt_deinterleave",
    "rt_wire_bytes", "rt_wire_pack", "rt_wire_unpack", "rt_debug_mesa_math", "rt_debug_shadow_tables", "rt_debug_predicted_classes",
    "rt_render_into_image", "rt_context_stream", "rt_mgpu_create", "rt_mgpu_destroy", "rt_mgpu_device_count", "rt_mgpu_set_scene", "rt_mgpu_set_noise",
    "rt_mgpu_set_skybox", "rt_mgpu_set_strip_rows", "rt_mgpu_render", "rt_mgpu_sync", "rt_mgpu_get_surfaces", "rt_mgpu_readback", "rt_mgpu_last_ms",
    "rt_mgpu_last_error",
]



class RtFrameDesc(ctypes.Structure):
    """``rt_frame_desc`` of include/rt_mi355.h."""
    _fields_ = [("enableAO", ctypes.c_int32), ("enableTAA", ctypes.c_int32), ("taaBlendFactor", ctypes.c_float),
                ("bloomThreshold", ctypes.c_float), ("bloomStrength", ctypes.c_float), ("bloomIterations", ctypes.c_int32),
                ("aoSamples", ctypes.POINTER(ctypes.c_float)), ("aoNoise", ctypes.POINTER(ctypes.c_float)),
                ("reserved", ctypes.c_float * 2)]


RT_OK = 0
STATUS_NAMES = {0: "RT_OK", -1: "RT_ERR_INVALID_ARG", -2: "RT_ERR_NO_DEVICE", -3: "RT_ERR_HIP",
                -4: "RT_ERR_TOO_LARGE", -5: "RT_ERR_NO_SURFACES", -6: "RT_ERR_PARSE"}


class RtError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {msg}")


def load_library(build_if_missing=True):
    """dlopen the in-tree librt_mi355.so (building it first if asked and absent)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("RT_LIB", _build.LIB_PATH)   # RT_LIB: experiment builds (tools/gpu_explore.py)
    if not os.path.exists(path):
        if not build_if_missing:
            raise RtError(-2, f"{path} not built (run python -m opengl_raytracing_amd.build)")
        _build.build_library()
    lib = ctypes.CDLL(path)
    if "RT_LIB" in os.environ:          # an experiment build (tools/gpu_try.py) may predate newer entry points: bind what it has
        class _Lenient:
            def __init__(self, real):
                object.__setattr__(self, "_real", real)

            def __getattr__(self, name):
                try:
                    return getattr(self._real, name)
                except AttributeError:
                    class _Missing:
                        argtypes = restype = None

                        def __call__(self, *a):
                            raise RtError(-2, f"{name} is not exported by {path}")
                    return _Missing()
        lib = _Lenient(lib)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    P = ctypes.POINTER
    lib.rt_create.argtypes = [P(vp), ci]
    lib.rt_destroy.argtypes = [vp]
    lib.rt_set_scene.argtypes = [vp, vp, ci, vp, ci]
    lib.rt_set_noise.argtypes = [vp, vp, ci, ci]
    lib.rt_set_skybox.argtypes = [vp, vp, ci]
    lib.rt_render.argtypes = [vp, P(L.RtParams)]
    lib.rt_render_to.argtypes = [vp, P(L.RtParams), vp, vp, vp, vp]
    lib.rt_sync.argtypes = [vp]
    lib.rt_readback.argtypes = [vp, vp, vp, vp]
    lib.rt_get_surfaces.argtypes = [vp, P(vp), P(vp), P(vp)]
    lib.rt_last_kernel_ms.argtypes = [vp, P(ctypes.c_float)]
    lib.rt_count_rays.argtypes = [vp, P(L.RtParams), P(ctypes.c_uint64)]
    lib.rt_count_rays_traced.argtypes = [vp, P(L.RtParams), P(ctypes.c_uint64)]
    lib.rt_set_variant.argtypes = [vp, ci]
    lib.rt_debug_stats.argtypes = [vp, P(ctypes.c_uint64)]
    lib.rt_debug_stats_ex.argtypes = [vp, P(ctypes.c_uint64)]
    lib.rt_debug_tile_costs.argtypes = [vp, P(ctypes.c_uint32), ci, P(ci), P(ci)]
    lib.rt_last_error.argtypes = [vp]
    lib.rt_last_error.restype = ctypes.c_char_p
    lib.rt_generate_aabb.argtypes = [vp, ci]
    lib.rt_camera_vectors.argtypes = [ctypes.c_float, ctypes.c_float, P(ctypes.c_float), P(ctypes.c_float), P(ctypes.c_float)]
    lib.rt_scene_parse.argtypes = [ctypes.c_char_p, vp, ci, P(ci), vp, ci, P(ci)]
    lib.rt_scene_write.argtypes = [vp, ci, vp, ci, vp, vp, ctypes.c_char_p, ctypes.c_size_t, P(ctypes.c_size_t)]
    cf = ctypes.c_float
    lib.rt_taa_resolve.argtypes = [vp, vp, vp, vp, vp, ci, ci, cf, cf, cf, vp]
    lib.rt_taa_jitter.argtypes = [ci, ci, ci, P(cf), P(cf)]
    lib.rt_bloom.argtypes = [vp, vp, vp, ci, ci, cf, cf, ci, vp]
    lib.rt_ssao.argtypes = [vp, vp, vp, vp, ci, ci, P(cf), ci, ci, P(cf), P(cf), P(cf), vp]
    lib.rt_ssao_blur.argtypes = [vp, vp, vp, ci, ci, ci, vp]
    lib.rt_camera_matrices.argtypes = [P(cf), P(cf), P(cf), cf, cf, P(cf), P(cf)]
    lib.rt_equirect_to_cubemap.argtypes = [vp, P(cf), ci, ci, ci, vp, ci]
    lib.rt_frame.argtypes = [vp, P(L.RtParams), P(RtFrameDesc), vp]
    lib.rt_frame_surfaces.argtypes = [vp, P(vp), P(vp), P(vp), P(vp), P(vp)]
    lib.rt_strip_local_rows.argtypes = [ci, ci, ci, ci]
    lib.rt_deinterleave.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ctypes.c_size_t, vp]
    lib.rt_wire_bytes.argtypes = [ctypes.c_size_t]
    lib.rt_wire_bytes.restype = ctypes.c_size_t
    lib.rt_wire_pack.argtypes = [vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    lib.rt_wire_unpack.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_size_t, vp, vp, vp, ci, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.rt_debug_mesa_math.argtypes = [vp, vp, ci]
    lib.rt_debug_shadow_tables.argtypes = [vp, vp, ctypes.c_size_t, P(ctypes.c_size_t), P(ci)]
    lib.rt_debug_predicted_classes.argtypes = [vp, vp, ci, P(ci)]
    lib.rt_render_into_image.argtypes = [vp, P(L.RtParams), vp, vp, vp, vp]
    lib.rt_context_stream.argtypes = [vp, P(vp)]
    lib.rt_mgpu_create.argtypes = [P(vp), P(ci), ci]
    lib.rt_mgpu_destroy.argtypes = [vp]
    lib.rt_mgpu_device_count.argtypes = [vp]
    lib.rt_mgpu_set_scene.argtypes = [vp, vp, ci, vp, ci]
    lib.rt_mgpu_set_noise.argtypes = [vp, vp, ci, ci]
    lib.rt_mgpu_set_skybox.argtypes = [vp, vp, ci]
    lib.rt_mgpu_set_strip_rows.argtypes = [vp, ci]
    lib.rt_mgpu_render.argtypes = [vp, P(L.RtParams)]
    lib.rt_mgpu_sync.argtypes = [vp]
    lib.rt_mgpu_get_surfaces.argtypes = [vp, P(vp), P(vp), P(vp), P(vp)]
    lib.rt_mgpu_readback.argtypes = [vp, vp, vp, vp]
    lib.rt_mgpu_last_ms.argtypes = [vp, P(ctypes.c_float), ci]
    lib.rt_mgpu_last_error.argtypes = [vp]
    lib.rt_mgpu_last_error.restype = ctypes.c_char_p
    for name in EXPORTS:
        if name not in ("rt_last_error", "rt_mgpu_last_error", "rt_wire_bytes"):
            getattr(lib, name).restype = ci
    _LIB = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


# ---- host-side feeders (no GPU) ------------------------------------------------------------
def generate_aabb(objects):
    """GenerateAABBForObject (/root/reference/src/SceneIO.h:75-104), in place."""
    assert objects.dtype == L.OBJECT_DTYPE and objects.flags["C_CONTIGUOUS"]
    rc = load_library().rt_generate_aabb(_ptr(objects), len(objects))
    if rc:
        raise RtError(rc, "rt_generate_aabb")
    return objects


def camera_vectors(yaw_deg=-90.0, pitch_deg=0.0):
    """Camera::UpdateVectors (/root/reference/src/Camera.h:26-34) -> (front, right, up)."""
    f, r, u = (ctypes.c_float * 3)(), (ctypes.c_float * 3)(), (ctypes.c_float * 3)()
    rc = load_library().rt_camera_vectors(yaw_deg, pitch_deg, f, r, u)
    if rc:
        raise RtError(rc, "rt_camera_vectors")
    return tuple(f), tuple(r), tuple(u)


def camera_matrices(position, front, up, fov_deg=45.0, aspect=16.0 / 9.0):
    """Camera::GetViewMatrix / GetProjectionMatrix (/root/reference/src/Camera.h:36-42) -> (view[16],
    projection[16]) float32, column-major."""
    a3 = lambda v: (ctypes.c_float * 3)(*[float(x) for x in v])
    view, proj = (ctypes.c_float * 16)(), (ctypes.c_float * 16)()
    rc = load_library().rt_camera_matrices(a3(position), a3(front), a3(up), fov_deg, aspect, view, proj)
    if rc:
        raise RtError(rc, "rt_camera_matrices")
    return np.array(view, dtype=np.float32), np.array(proj, dtype=np.float32)


def ssao_kernel(seed=0x55A0):
    """The 64 hemisphere samples and the 4x4 rotation texture of AOManager::InitSSAO (AO.cpp:23-51), same
    construction; the reference draws them from std::default_random_engine (implementation-defined
    sequence), this generator uses SplitMix64 -- they are inputs of rt_ssao either way."""
    from .scenes import SplitMix64
    rng = SplitMix64(seed)
    samples = np.zeros((64, 3), dtype=np.float32)
    for i in range(64):
        s = np.array([rng.uniform(0, 1) * 2.0 - 1.0, rng.uniform(0, 1) * 2.0 - 1.0, rng.uniform(0, 1)], dtype=np.float32)
        s = s / np.float32(np.sqrt(np.dot(s, s)))
        s = s * np.float32(rng.uniform(0, 1))
        scale = np.float32(i) / np.float32(64.0)
        scale = np.float32(0.1) + (scale * scale) * np.float32(0.9)
        samples[i] = s * scale
    noise = np.zeros((4, 4, 4), dtype=np.float32)      # uploaded as GL_RGB into RGBA32F: alpha reads 1
    for k in range(16):
        noise[k // 4, k % 4, 0] = rng.uniform(0, 1) * 2.0 - 1.0
        noise[k // 4, k % 4, 1] = rng.uniform(0, 1) * 2.0 - 1.0
    noise[..., 3] = 1.0
    return samples, noise


def parse_scene(text, max_objects=512, max_lights=64):
    """SceneIO::Load (/root/reference/src/SceneIO.h:108-122) on in-memory text."""
    objs = np.zeros(max_objects, dtype=L.OBJECT_DTYPE)
    lts = np.zeros(max_lights, dtype=L.LIGHT_DTYPE)
    no, nl = ctypes.c_int(0), ctypes.c_int(0)
    rc = load_library().rt_scene_parse(text.encode("utf-8"), _ptr(objs), max_objects, ctypes.byref(no),
                                       _ptr(lts), max_lights, ctypes.byref(nl))
    if rc:
        raise RtError(rc, "rt_scene_parse")
    return objs[: no.value].copy(), lts[: nl.value].copy()


def write_scene(objects, lights):
    """SceneIO::Save (/root/reference/src/SceneIO.h:124-142) -> text."""
    lib = load_library()
    objects = np.ascontiguousarray(objects)
    lights = np.ascontiguousarray(lights)
    need = ctypes.c_size_t(0)
    rc = lib.rt_scene_write(_ptr(objects), len(objects), _ptr(lights), len(lights), None, None, None, 0, ctypes.byref(need))
    if rc:
        raise RtError(rc, "rt_scene_write")
    buf = ctypes.create_string_buffer(need.value)
    rc = lib.rt_scene_write(_ptr(objects), len(objects), _ptr(lights), len(lights), None, None, buf, need.value, ctypes.byref(need))
    if rc:
        raise RtError(rc, "rt_scene_write")
    return buf.value.decode()


def taa_jitter(frame_count, width, height):
    """uJitterX/uJitterY of /root/reference/src/ForwardShadingPipeline.cpp:241-242."""
    jx, jy = ctypes.c_float(), ctypes.c_float()
    rc = load_library().rt_taa_jitter(frame_count, width, height, ctypes.byref(jx), ctypes.byref(jy))
    if rc:
        raise RtError(rc, "rt_taa_jitter")
    return jx.value, jy.value


def mesa_math(x):
    """(sin, cos, tan, exp) of float32 x as the HIP path's host side evaluates them (csrc/rt_mesa_math.h)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros((len(x), 4), dtype=np.float32)
    rc = load_library().rt_debug_mesa_math(_ptr(x), _ptr(out), len(x))
    if rc:
        raise RtError(rc, "rt_debug_mesa_math")
    return out


def strip_local_rows(height, strip_rows, strip_count, strip_index):
    n = load_library().rt_strip_local_rows(height, strip_rows, strip_count, strip_index)
    if n < 0:
        raise RtError(n, "rt_strip_local_rows")
    return n


# ---- the device context ---------------------------------------------------------------------
class RayTracer:
    def __init__(self, device=0):
        self.lib = load_library()
        self.ctx = ctypes.c_void_p()
        rc = self.lib.rt_create(ctypes.byref(self.ctx), device)
        if rc:
            self.ctx = None
            raise RtError(rc, "rt_create (is a HIP device present?)")
        self._region = None
        v = int(os.environ.get("RT_VARIANT", "-1"))   # A/B switch for measurements and tests
        if v >= 0:
            self.set_variant(v)

    def _check(self, rc, what):
        if rc:
            raise RtError(rc, f"{what}: {self.lib.rt_last_error(self.ctx).decode()}")

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.rt_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_scene(self, objects, lights):
        objects = np.ascontiguousarray(objects)
        lights = np.ascontiguousarray(lights)
        assert objects.dtype.itemsize == L.OBJECT_STRIDE and lights.dtype.itemsize == L.LIGHT_STRIDE
        self._check(self.lib.rt_set_scene(self.ctx, _ptr(objects) if len(objects) else None, len(objects),
                                          _ptr(lights) if len(lights) else None, len(lights)), "rt_set_scene")

    def set_noise(self, r8):
        if r8 is None:
            self._check(self.lib.rt_set_noise(self.ctx, None, 0, 0), "rt_set_noise")
            return
        r8 = np.ascontiguousarray(r8, dtype=np.uint8)
        self._check(self.lib.rt_set_noise(self.ctx, _ptr(r8), r8.shape[1], r8.shape[0]), "rt_set_noise")

    def set_skybox(self, faces):
        if faces is None:
            self._check(self.lib.rt_set_skybox(self.ctx, None, 0), "rt_set_skybox")
            return
        faces = np.ascontiguousarray(faces, dtype=np.float16)
        assert faces.ndim == 4 and faces.shape[0] == 6 and faces.shape[1] == faces.shape[2] and faces.shape[3] == 3
        self._check(self.lib.rt_set_skybox(self.ctx, _ptr(faces), faces.shape[1]), "rt_set_skybox")

    def load(self, scene):
        """Upload a scenes.Scene (objects, lights, noise, skybox)."""
        self.set_scene(scene.objects, scene.lights)
        self.set_noise(scene.noise)
        self.set_skybox(scene.skybox if scene.use_skybox else None)

    def set_variant(self, v):
        self._check(self.lib.rt_set_variant(self.ctx, int(v)), "rt_set_variant")

    def render(self, params):
        self._check(self.lib.rt_render(self.ctx, ctypes.byref(params)), "rt_render")
        self._region = (params.regionW, params.regionH)

    def render_to(self, params, d_color, d_position, d_normal, stream=None):
        """Render into caller-owned device memory (raw device pointers as ints)."""
        self._check(self.lib.rt_render_to(self.ctx, ctypes.byref(params), ctypes.c_void_p(d_color),
                                          ctypes.c_void_p(d_position), ctypes.c_void_p(d_normal),
                                          ctypes.c_void_p(stream) if stream else None), "rt_render_to")

    def sync(self):
        self._check(self.lib.rt_sync(self.ctx), "rt_sync")

    def readback(self):
        """-> (gColor float32[h,w,4], gPosition float32[h,w,4], gNormal float16[h,w,4]); row 0 = bottom."""
        w, h = self._region
        col = np.empty((h, w, 4), dtype=np.float32)
        pos = np.empty((h, w, 4), dtype=np.float32)
        nrm = np.empty((h, w, 4), dtype=np.float16)
        self._check(self.lib.rt_readback(self.ctx, _ptr(col), _ptr(pos), _ptr(nrm)), "rt_readback")
        return col, pos, nrm

    def get_surfaces(self):
        """Device pointers (ints) of the context-owned gColor, gPosition, gNormal of the last rt_render."""
        dc, dp, dn = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self.lib.rt_get_surfaces(self.ctx, ctypes.byref(dc), ctypes.byref(dp), ctypes.byref(dn)), "rt_get_surfaces")
        return dc.value, dp.value, dn.value

    def last_kernel_ms(self):
        ms = ctypes.c_float()
        self._check(self.lib.rt_last_kernel_ms(self.ctx, ctypes.byref(ms)), "rt_last_kernel_ms")
        return ms.value

    def count_rays(self, params):
        n = ctypes.c_uint64()
        self._check(self.lib.rt_count_rays(self.ctx, ctypes.byref(params), ctypes.byref(n)), "rt_count_rays")
        return n.value

    def count_rays_traced(self, params):
        """Rays the production kernel traverses (keeps its dead-ray skips); <= count_rays."""
        n = ctypes.c_uint64()
        self._check(self.lib.rt_count_rays_traced(self.ctx, ctypes.byref(params), ctypes.byref(n)), "rt_count_rays_traced")
        return n.value

    def taa_resolve(self, d_current, d_history, d_normal, d_out, width, height, blend, jx, jy, stream=None):
        """TAA resolve pass on device surfaces (raw device pointers as ints)."""
        self._check(self.lib.rt_taa_resolve(self.ctx, ctypes.c_void_p(d_current), ctypes.c_void_p(d_history),
                                            ctypes.c_void_p(d_normal), ctypes.c_void_p(d_out), width, height, blend, jx, jy,
                                            ctypes.c_void_p(stream) if stream else None), "rt_taa_resolve")

    def bloom(self, d_scene, d_out, width, height, threshold=1.0, strength=0.5, iterations=10, stream=None):
        """Bloom chain on device surfaces (raw device pointers as ints)."""
        self._check(self.lib.rt_bloom(self.ctx, ctypes.c_void_p(d_scene), ctypes.c_void_p(d_out), width, height, threshold,
                                      strength, iterations, ctypes.c_void_p(stream) if stream else None), "rt_bloom")

    def tile_costs(self):
        """(costs[tilesY, tilesX] uint32, cycles/64 per tile) of the last feedback-scheduled launch."""
        import numpy as _np
        n, tx = ctypes.c_int(0), ctypes.c_int(0)
        probe = (ctypes.c_uint32 * 1)()
        self._check(self.lib.rt_debug_tile_costs(self.ctx, probe, 0, ctypes.byref(n), ctypes.byref(tx)), "rt_debug_tile_costs")
        if n.value == 0:
            return _np.zeros((0, 0), dtype=_np.uint32)
        out = (ctypes.c_uint32 * n.value)()
        self._check(self.lib.rt_debug_tile_costs(self.ctx, out, n.value, ctypes.byref(n), ctypes.byref(tx)), "rt_debug_tile_costs")
        return _np.frombuffer(out, dtype=_np.uint32).reshape(-1, tx.value).copy()

    def frame(self, params, enable_ao=True, enable_taa=True, taa_blend=0.1, bloom_threshold=1.0, bloom_strength=0.5,
              bloom_iterations=10, ao_samples=None, ao_noise=None, d_display=None):
        """One iteration of the reference's Render() GPU work (ray trace, AO, bloom, TAA) on the context's surfaces."""
        d = RtFrameDesc()
        d.enableAO, d.enableTAA, d.taaBlendFactor = int(bool(enable_ao)), int(bool(enable_taa)), taa_blend
        d.bloomThreshold, d.bloomStrength, d.bloomIterations = bloom_threshold, bloom_strength, bloom_iterations
        keep = []
        if enable_ao:
            if ao_samples is None or ao_noise is None:
                ao_samples, ao_noise = ssao_kernel()
            for name, arr, n in (("aoSamples", ao_samples, 192), ("aoNoise", ao_noise, 64)):
                a = np.ascontiguousarray(arr, dtype=np.float32).reshape(n)
                keep.append(a)
                setattr(d, name, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        self._check(self.lib.rt_frame(self.ctx, ctypes.byref(params), ctypes.byref(d),
                                      ctypes.c_void_p(d_display) if d_display else None), "rt_frame")

    def frame_surfaces(self):
        """(dColor, dPosition, dNormal, dAO, dHistory) device pointers (ints or None) after rt_frame."""
        ptrs = [ctypes.c_void_p() for _ in range(5)]
        self._check(self.lib.rt_frame_surfaces(self.ctx, *[ctypes.byref(q) for q in ptrs]), "rt_frame_surfaces")
        return tuple(q.value for q in ptrs)

    def equirect_to_cubemap(self, equirect_rgb, size, d_faces_out=None, install=False):
        """ConvertHDRToCubemap: equirect f32[h,w,3] (row 0 = bottom) -> six RGB16F faces on the device
        (d_faces_out: raw pointer, 6*size*size*3 halfs) and / or installed as this context's skybox."""
        e = np.ascontiguousarray(equirect_rgb, dtype=np.float32)
        h, w = e.shape[:2]
        self._check(self.lib.rt_equirect_to_cubemap(self.ctx, e.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), w, h, size,
                                                    ctypes.c_void_p(d_faces_out) if d_faces_out else None, int(bool(install))),
                    "rt_equirect_to_cubemap")

    def ssao(self, d_position, d_normal, d_out, width, height, noise, samples, projection, view, stream=None):
        """SSAO on the G-buffer surfaces (raw device pointers as ints); noise [nh,nw,4], samples [64,3],
        matrices [16] are host arrays."""
        fa = lambda x, n: np.ascontiguousarray(x, dtype=np.float32).reshape(n)
        noise = np.ascontiguousarray(noise, dtype=np.float32)
        nh, nw = noise.shape[:2]
        fp = lambda x: x.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        nz, sm, pj, vw = fa(noise, nh * nw * 4), fa(samples, 192), fa(projection, 16), fa(view, 16)
        self._check(self.lib.rt_ssao(self.ctx, ctypes.c_void_p(d_position), ctypes.c_void_p(d_normal), ctypes.c_void_p(d_out),
                                     width, height, fp(nz), nw, nh, fp(sm), fp(pj), fp(vw),
                                     ctypes.c_void_p(stream) if stream else None), "rt_ssao")

    def ssao_blur(self, d_in, d_out, width, height, horizontal=False, stream=None):
        self._check(self.lib.rt_ssao_blur(self.ctx, ctypes.c_void_p(d_in), ctypes.c_void_p(d_out), width, height,
                                          int(bool(horizontal)), ctypes.c_void_p(stream) if stream else None), "rt_ssao_blur")

    def predicted_classes(self, n_tiles):
        """Cost classes (uint8[n_tiles], raster tile order) of the last predicted launch."""
        out = np.zeros(n_tiles, dtype=np.uint8)
        n = ctypes.c_int(0)
        self._check(self.lib.rt_debug_predicted_classes(self.ctx, _ptr(out), n_tiles, ctypes.byref(n)), "rt_debug_predicted_classes")
        return out

    def shadow_tables(self):
        """(dwords uint32[n], words per cell) of the current scene's shadow tables (headers + cells), or (None, 0)."""
        n, w = ctypes.c_size_t(0), ctypes.c_int(0)
        self._check(self.lib.rt_debug_shadow_tables(self.ctx, None, 0, ctypes.byref(n), ctypes.byref(w)), "rt_debug_shadow_tables")
        if n.value == 0:
            return None, 0
        out = np.zeros(n.value, dtype=np.uint32)
        self._check(self.lib.rt_debug_shadow_tables(self.ctx, _ptr(out), n.value, ctypes.byref(n), ctypes.byref(w)), "rt_debug_shadow_tables")
        return out, w.value

    def debug_stats(self):
        out = (ctypes.c_uint64 * 4)()
        self._check(self.lib.rt_debug_stats(self.ctx, out), "rt_debug_stats")
        return list(out)

    def debug_stats_ex(self):
        out = (ctypes.c_uint64 * 32)()
        self._check(self.lib.rt_debug_stats_ex(self.ctx, out), "rt_debug_stats_ex")
        return list(out)

    def deinterleave(self, d_src, d_dst, width, height, bytes_per_pixel, strip_rows, strip_count,
                     rank_stride_bytes, stream=None):
        self._check(self.lib.rt_deinterleave(self.ctx, ctypes.c_void_p(d_src), ctypes.c_void_p(d_dst), width,
                                             height, bytes_per_pixel, strip_rows, strip_count, rank_stride_bytes,
                                             ctypes.c_void_p(stream) if stream else None), "rt_deinterleave")

    def wire_pack(self, d_color, d_pos, d_normal, d_wire, n_pixels, stream=None):
        """This rank's three surfaces -> the 30 B/pixel gather wire format (device pointers as ints)."""
        self._check(self.lib.rt_wire_pack(self.ctx, ctypes.c_void_p(d_color), ctypes.c_void_p(d_pos), ctypes.c_void_p(d_normal),
                                          ctypes.c_void_p(d_wire), n_pixels, ctypes.c_void_p(stream) if stream else None),
                    "rt_wire_pack")

    def wire_unpack(self, d_wire, rank_stride_bytes, rank_pixels, d_color, d_pos, d_normal, width, height, strip_rows,
                    strip_count, root=None, root_strips=1, stream=None):
        """Gathered wire buffers -> full rgba surfaces in image order (alpha restored to 1.0).  root = (d_color,
        d_pos, d_normal) of rank 0's own local surfaces: its rows are copied from there instead of wire slot 0."""
        r = [ctypes.c_void_p(x) for x in root] if root else [None, None, None]
        self._check(self.lib.rt_wire_unpack(self.ctx, ctypes.c_void_p(d_wire), rank_stride_bytes, rank_pixels, r[0], r[1], r[2],
                                            root_strips, ctypes.c_void_p(d_color), ctypes.c_void_p(d_pos),
                                            ctypes.c_void_p(d_normal), width, height, strip_rows, strip_count,
                                            ctypes.c_void_p(stream) if stream else None), "rt_wire_unpack")


class MultiGpuRayTracer:
    """rt_mgpu_*: one frame on N devices from one process; the devices' kernels store their strips straight into device 0's frame
    (include/rt_mi355.h).  `devices` may repeat an id (the N-way plan on one GPU)."""

    def __init__(self, devices, strip_rows=8):
        self.lib = load_library()
        self.m = ctypes.c_void_p()
        arr = (ctypes.c_int * len(devices))(*devices)
        rc = self.lib.rt_mgpu_create(ctypes.byref(self.m), arr, len(devices))
        if rc:
            self.m = None
            raise RtError(rc, "rt_mgpu_create (peer access between the devices?)")
        self.n = len(devices)
        self._check(self.lib.rt_mgpu_set_strip_rows(self.m, strip_rows), "rt_mgpu_set_strip_rows")
        self._size = None

    def _check(self, rc, what):
        if rc:
            raise RtError(rc, f"{what}: {self.lib.rt_mgpu_last_error(self.m).decode()}")

    def close(self):
        if getattr(self, "m", None):
            self.lib.rt_mgpu_destroy(self.m)
            self.m = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load(self, scene):
        objects, lights = np.ascontiguousarray(scene.objects), np.ascontiguousarray(scene.lights)
        self._check(self.lib.rt_mgpu_set_scene(self.m, _ptr(objects) if len(objects) else None, len(objects),
                                               _ptr(lights) if len(lights) else None, len(lights)), "rt_mgpu_set_scene")
        if scene.noise is None:
            self._check(self.lib.rt_mgpu_set_noise(self.m, None, 0, 0), "rt_mgpu_set_noise")
        else:
            r8 = np.ascontiguousarray(scene.noise, dtype=np.uint8)
            self._check(self.lib.rt_mgpu_set_noise(self.m, _ptr(r8), r8.shape[1], r8.shape[0]), "rt_mgpu_set_noise")
        faces = np.ascontiguousarray(scene.skybox, dtype=np.float16) if scene.use_skybox else None
        self._check(self.lib.rt_mgpu_set_skybox(self.m, _ptr(faces), faces.shape[1] if faces is not None else 0), "rt_mgpu_set_skybox")

    def render(self, params):
        self._check(self.lib.rt_mgpu_render(self.m, ctypes.byref(params)), "rt_mgpu_render")
        self._size = (params.width, params.height)

    def sync(self):
        self._check(self.lib.rt_mgpu_sync(self.m), "rt_mgpu_sync")

    def readback(self):
        w, h = self._size
        col = np.empty((h, w, 4), dtype=np.float32)
        pos = np.empty((h, w, 4), dtype=np.float32)
        nrm = np.empty((h, w, 4), dtype=np.float16)
        self._check(self.lib.rt_mgpu_readback(self.m, _ptr(col), _ptr(pos), _ptr(nrm)), "rt_mgpu_readback")
        return col, pos, nrm

    def last_ms(self):
        out = (ctypes.c_float * self.n)()
        self._check(self.lib.rt_mgpu_last_ms(self.m, out, self.n), "rt_mgpu_last_ms")
        return list(out)
