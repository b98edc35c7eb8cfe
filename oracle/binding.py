"""TEST INFRASTRUCTURE: ctypes binding of oracle/liboracle_rt.so (the CPU restatement) and a
runner for oracle/_ref/gl_harness (the reference's own GLSL on Mesa llvmpipe).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (opengl_raytracing_amd/) never imports this module.
"""
import ctypes
import json
import os
import struct
import subprocess
import tempfile

import numpy as np

ORACLE_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle_rt.so")
HARNESS = os.path.join(ORACLE_DIR, "_ref", "gl_harness")
REFERENCE_GLSL = "/root/reference/shader/raytracingCs.glsl"

_LIB = None


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            subprocess.run(["make", "-C", ORACLE_DIR, "liboracle_rt.so"], check=True)
        lib = ctypes.CDLL(LIB_PATH)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        lib.orc_render.argtypes = [vp, ci, vp, ci, vp, vp, ci, ci, vp, ci, vp, vp, vp, ctypes.POINTER(ctypes.c_uint64), ci]
        lib.orc_render.restype = ci
        lib.orc_generate_aabb.argtypes = [vp, ci]
        lib.orc_generate_aabb.restype = None
        lib.orc_halton.argtypes = [vp, vp, vp, ci]
        lib.orc_float_to_half_rtz.argtypes = [vp, vp, ci]
        lib.orc_sample_cube.argtypes = [vp, ci, vp, vp, ci]
        cf = ctypes.c_float
        lib.orc_taa_resolve.argtypes = [vp, vp, vp, ci, ci, cf, cf, cf, vp]
        lib.orc_taa_resolve.restype = None
        lib.orc_taa_jitter.argtypes = [ci, ci, ci, ctypes.POINTER(cf), ctypes.POINTER(cf)]
        lib.orc_taa_jitter.restype = None
        _LIB = lib
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def render(scene, params, nthreads=0):
    """CPU restatement of one dispatch.  -> (gColor f32[h,w,4], gPosition f32[h,w,4],
    gNormal f16[h,w,4], rays)."""
    lib = load()
    w, h = params.regionW, params.regionH
    col = np.zeros((h, w, 4), dtype=np.float32)
    pos = np.zeros((h, w, 4), dtype=np.float32)
    nrm = np.zeros((h, w, 4), dtype=np.float16)
    rays = ctypes.c_uint64(0)
    objs = np.ascontiguousarray(scene.objects)
    lts = np.ascontiguousarray(scene.lights)
    noise = np.ascontiguousarray(scene.noise) if scene.noise is not None else None
    sky = np.ascontiguousarray(scene.skybox) if (scene.skybox is not None and scene.use_skybox) else None
    # orc_params.reserved0 carries the oracle's DIAGNOSTIC bits (rt_oracle.h: 1 = llvmpipe's polynomial pow, 2 / 4 = its loop
    # limiter); every parity test against the HIP path leaves it 0 -- anything else is refused rather than silently honoured
    if not 0 <= int(params.reserved0) <= 7:
        raise ValueError(f"params.reserved0 = {params.reserved0}: not a set of oracle diagnostic bits")
    rc = lib.orc_render(_ptr(objs), len(objs), _ptr(lts), len(lts), ctypes.byref(params),
                        _ptr(noise), noise.shape[1] if noise is not None else 0,
                        noise.shape[0] if noise is not None else 0,
                        _ptr(sky), sky.shape[1] if sky is not None else 0,
                        _ptr(col), _ptr(pos), _ptr(nrm), ctypes.byref(rays), nthreads)
    if rc:
        raise RuntimeError(f"orc_render failed: {rc}")
    return col, pos, nrm, rays.value


def generate_aabb(objects):
    load().orc_generate_aabb(_ptr(objects), len(objects))
    return objects


def halton(index, base):
    index = np.ascontiguousarray(index, dtype=np.int32)
    base = np.ascontiguousarray(base, dtype=np.int32)
    out = np.zeros(len(index), dtype=np.float32)
    load().orc_halton(_ptr(index), _ptr(base), _ptr(out), len(index))
    return out


def float_to_half_rtz(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(x.shape, dtype=np.uint16)
    load().orc_float_to_half_rtz(_ptr(x), _ptr(out), x.size)
    return out


def sample_cube(sky, dirs):
    sky = np.ascontiguousarray(sky, dtype=np.float16)
    dirs = np.ascontiguousarray(dirs, dtype=np.float32)
    out = np.zeros_like(dirs)
    load().orc_sample_cube(_ptr(sky), sky.shape[1], _ptr(dirs), _ptr(out), len(dirs))
    return out


def taa_resolve(current, history, normal, blend, jx, jy):
    """CPU restatement of taaFs.glsl.  current/history f32[h,w,4], normal f16|f32[h,w,4] -> f32[h,w,4]."""
    current = np.ascontiguousarray(current, dtype=np.float32)
    history = np.ascontiguousarray(history, dtype=np.float32)
    normal = np.ascontiguousarray(normal.astype(np.float32))
    h, w = current.shape[:2]
    out = np.zeros((h, w, 4), dtype=np.float32)
    load().orc_taa_resolve(_ptr(current), _ptr(history), _ptr(normal), w, h, blend, jx, jy, _ptr(out))
    return out


def taa_jitter(frame_count, w, h):
    jx, jy = ctypes.c_float(), ctypes.c_float()
    load().orc_taa_jitter(frame_count, w, h, ctypes.byref(jx), ctypes.byref(jy))
    return jx.value, jy.value


# ---- the reference shader itself on Mesa llvmpipe ------------------------------------------
def harness_available():
    return os.path.exists(HARNESS) and os.path.exists(REFERENCE_GLSL)


def write_job(path, scene, params):
    """RTJOB1 file for gl_harness (layout: oracle/gl_harness.c JobHeader)."""
    noise = scene.noise
    sky = scene.skybox if (scene.skybox is not None and scene.use_skybox) else None
    hdr = struct.pack(
        "<8s16i20f", b"RTJOB1\0\0", params.width, params.height, len(scene.objects), len(scene.lights),
        params.maxRayDepth, params.frameCount, params.useSkybox,
        noise.shape[1] if noise is not None else 0, noise.shape[0] if noise is not None else 0,
        sky.shape[1] if sky is not None else 0, 0, 0, 0, 0, 0, 0,
        *params.camPos, *params.camDir, *params.camUp, *params.camRight,
        params.fovDeg, params.focalLength, params.maxRayDistance, *params.noiseScale, 0.0, 0.0, 0.0)
    assert len(hdr) == 152
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(np.ascontiguousarray(scene.objects).tobytes())
        f.write(np.ascontiguousarray(scene.lights).tobytes())
        if noise is not None:
            f.write(np.ascontiguousarray(noise, dtype=np.uint8).tobytes())
        if sky is not None:
            f.write(np.ascontiguousarray(sky, dtype=np.float16).tobytes())


def run_reference(scene, params, repeat=0, shipped_dispatch=False, threads=None, glsl=REFERENCE_GLSL, groups=None, crop=None):
    """Run the reference's GLSL unmodified.  GL compute has no dispatch offset, so the default is the
    full frame; `groups=(gx, gy)` dispatches only the bottom-left gx x gy workgroups of 32x32 pixels of
    the full-size images (same invocations as in the full dispatch), `crop=(x0, y0, w, h)` returns only
    that window.  -> (gColor, gPosition, gNormal-as-f32, info dict)."""
    if not harness_available():
        raise RuntimeError("gl_harness or the reference GLSL is not available here")
    w, h = params.width, params.height
    with tempfile.TemporaryDirectory(prefix="rtjob_") as d:
        job = os.path.join(d, "job.bin")
        write_job(job, scene, params)
        cmd = [HARNESS, "render", glsl, job, os.path.join(d, "out"), "--repeat", str(repeat)]
        if shipped_dispatch:
            cmd.append("--shipped-dispatch")
        if groups:
            cmd += ["--groups", str(groups[0]), str(groups[1])]
        if crop:
            cmd += ["--crop"] + [str(int(v)) for v in crop]
            w, h = int(crop[2]), int(crop[3])
        env = dict(os.environ)
        if threads:
            env["LP_NUM_THREADS"] = str(threads)
        out = subprocess.run(cmd, check=True, capture_output=True, text=True, env=env)
        info = json.loads(out.stdout.strip().splitlines()[-1])
        res = [np.fromfile(os.path.join(d, f"out.{s}.f32"), dtype=np.float32).reshape(h, w, 4)
               for s in ("color", "pos", "normal")]
    return res[0], res[1], res[2], info


def run_probe(glsl_text, in_array, out_dtype, out_count, groups):
    """Micro-kernel mode: SSBO 0 = in_array bytes, SSBO 1 = out_count items of out_dtype."""
    with tempfile.TemporaryDirectory(prefix="rtprobe_") as d:
        g, i, o = (os.path.join(d, n) for n in ("p.glsl", "in.bin", "out.bin"))
        with open(g, "w") as f:
            f.write(glsl_text)
        np.ascontiguousarray(in_array).tofile(i)
        nbytes = int(np.dtype(out_dtype).itemsize * out_count)
        subprocess.run([HARNESS, "probe", g, i, o, str(nbytes), str(groups)], check=True, capture_output=True)
        return np.fromfile(o, dtype=out_dtype)


def run_postfx(vs_path, fs_path, out_w, out_h, textures, uniforms, out_half=False, cube=False):
    """One full-screen fragment pass of the reference (gl_harness postfx).
    textures: list of (sampler_name, float32 array [h,w,4], dict(half=False, linear=False, clamp=False));
    uniforms: list of (name, value) with value int | float | tuple.  -> float32 [out_h, out_w, 4]."""
    if not (os.path.exists(HARNESS) and os.path.exists(vs_path) and os.path.exists(fs_path)):
        raise RuntimeError("gl_harness or the reference shaders are not available here")
    with tempfile.TemporaryDirectory(prefix="rtpfx_") as d:
        job, out = os.path.join(d, "job.bin"), os.path.join(d, "out.f32")
        with open(job, "wb") as f:
            f.write(b"PFXJOB1\0")
            n_records = sum(4 if (not isinstance(v, (bool, int, float, np.integer, np.floating)) and len(v) == 16) else 1
                            for _, v in uniforms)
            f.write(struct.pack("<5i", out_w, out_h, int(out_half) | (2 if cube else 0), len(textures), n_records))
            for name, arr, opt in textures:
                arr = np.ascontiguousarray(arr, dtype=np.float32)
                h, w = arr.shape[:2]
                f.write(struct.pack("<32s5i", name.encode(), w, h, int(opt.get("half", False)), int(opt.get("linear", False)),
                                    int(opt.get("clamp", False))))
                f.write(arr.tobytes())
            for name, val in uniforms:
                if isinstance(val, (bool, int, np.integer)):
                    f.write(struct.pack("<32sii3f", name.encode(), 0, int(val), 0.0, 0.0, 0.0))
                elif isinstance(val, (float, np.floating)):
                    f.write(struct.pack("<32si4f", name.encode(), 1, float(val), 0.0, 0.0, 0.0))
                elif len(val) == 16:          # mat4, column-major: four column records
                    for c in range(4):
                        f.write(struct.pack("<32si4f", name.encode(), 10 + c, *[float(x) for x in val[4 * c:4 * c + 4]]))
                else:
                    v = [float(x) for x in val] + [0.0] * (4 - len(val))
                    f.write(struct.pack("<32si4f", name.encode(), {2: 2, 3: 3}.get(len(val), 4), *v))
        subprocess.run([HARNESS, "postfx", vs_path, fs_path, job, out], check=True, capture_output=True)
        return np.fromfile(out, dtype=np.float32).reshape(out_h, out_w, 4)


def bloom(scene, threshold=1.0, strength=0.5, iterations=10, keep=False):
    """CPU restatement of the reference's bloom chain (extract, `iterations` alternating blurs starting
    horizontal, combine).  scene f32[h,w,4] -> combined f32[h,w,4] (and the intermediate half
    textures when keep=True)."""
    lib = load()
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib.orc_bloom_extract.argtypes = [vp, ci, ci, cf, vp]
    lib.orc_bloom_blur.argtypes = [vp, ci, ci, ci, vp]
    lib.orc_bloom_combine.argtypes = [vp, vp, ci, ci, cf, vp]
    scene = np.ascontiguousarray(scene, dtype=np.float32)
    h, w = scene.shape[:2]
    a = np.zeros((h, w, 4), dtype=np.uint16)
    b = np.zeros_like(a)
    lib.orc_bloom_extract(_ptr(scene), w, h, threshold, _ptr(a))
    stages = [a.copy()]
    horizontal = 1
    for _ in range(iterations):
        lib.orc_bloom_blur(_ptr(a), w, h, horizontal, _ptr(b))
        a, b = b, a
        horizontal ^= 1
        if keep:
            stages.append(a.copy())
    out = np.zeros((h, w, 4), dtype=np.float32)
    lib.orc_bloom_combine(_ptr(scene), _ptr(a), w, h, strength, _ptr(out))
    return (out, [s.view(np.float16) for s in stages]) if keep else out


def ssao(position, normal, noise, samples, projection, view):
    """CPU restatement of ssaoFs.glsl.  position f32[h,w,4], normal f32/f16[h,w,4], noise f32[nh,nw,4],
    samples f32[64,3], projection/view f32[16] column-major -> f32[h,w]."""
    lib = load()
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.orc_ssao.argtypes = [vp, vp, ci, ci, vp, ci, ci, vp, vp, vp, vp]
    position = np.ascontiguousarray(position, dtype=np.float32)
    normal = np.ascontiguousarray(np.asarray(normal).astype(np.float32))
    noise = np.ascontiguousarray(noise, dtype=np.float32)
    samples = np.ascontiguousarray(samples, dtype=np.float32)
    projection = np.ascontiguousarray(projection, dtype=np.float32).ravel()
    view = np.ascontiguousarray(view, dtype=np.float32).ravel()
    h, w = position.shape[:2]
    out = np.zeros((h, w), dtype=np.float32)
    lib.orc_ssao(_ptr(position), _ptr(normal), w, h, _ptr(noise), noise.shape[1], noise.shape[0], _ptr(samples),
                 _ptr(projection), _ptr(view), _ptr(out))
    return out


def ssao_blur(ao, horizontal=False):
    """CPU restatement of ssao_blurFs.glsl (one direction).  ao f32[h,w] -> f32[h,w]."""
    lib = load()
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.orc_ssao_blur.argtypes = [vp, ci, ci, ci, vp]
    ao = np.ascontiguousarray(ao, dtype=np.float32)
    h, w = ao.shape
    out = np.zeros_like(ao)
    lib.orc_ssao_blur(_ptr(ao), w, h, int(bool(horizontal)), _ptr(out))
    return out


def equirect_to_cubemap(equirect, size):
    """CPU restatement of ConvertHDRToCubemap + skyboxFs.glsl.  equirect f32[h,w,3] (fp16-representable values,
    row 0 = bottom) -> faces f16[6,size,size,3]."""
    lib = load()
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.orc_equirect_to_cubemap.argtypes = [vp, ci, ci, ci, vp]
    equirect = np.ascontiguousarray(equirect, dtype=np.float32)
    h, w = equirect.shape[:2]
    faces = np.zeros((6, size, size, 3), dtype=np.uint16)
    lib.orc_equirect_to_cubemap(_ptr(equirect), w, h, size, _ptr(faces))
    return faces.view(np.float16)


def mesa_atan2_asin(y, x, z):
    lib = load()
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.orc_mesa_atan2_asin.argtypes = [vp, vp, vp, ci, vp, vp]
    y, x, z = (np.ascontiguousarray(a, dtype=np.float32) for a in (y, x, z))
    a, s = np.zeros_like(y), np.zeros_like(y)
    lib.orc_mesa_atan2_asin(_ptr(y), _ptr(x), _ptr(z), len(y), _ptr(a), _ptr(s))
    return a, s


def mesa_trig(x):
    """(sin, cos, tan) of float32 x as oracle/rt_oracle.c's mesa_sinf / mesa_cosf / mesa_tanf evaluate them."""
    lib = load()
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.orc_mesa_trig.argtypes = [vp, vp, ci]
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros((len(x), 3), dtype=np.float32)
    lib.orc_mesa_trig(_ptr(x), _ptr(out), len(x))
    return out


def mesa_explog(x):
    """(log2, exp2, pow(x,5), exp) of float32 x as the oracle's mesa_* restatements evaluate them."""
    lib = load()
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.orc_mesa_explog.argtypes = [vp, vp, ci]
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros((len(x), 4), dtype=np.float32)
    lib.orc_mesa_explog(_ptr(x), _ptr(out), len(x))
    return out
