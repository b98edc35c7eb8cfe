#!/usr/bin/env python3
"""Generate the committed golden fixtures by running the REFERENCE's own shader
(/root/reference/shader/raytracingCs.glsl, read at run time, never copied) unmodified on
Mesa llvmpipe through oracle/_ref/gl_harness.

Runs only in the build container (needs /root/reference and Mesa's swrast_dri.so); the GPU box
and the CPU test-suite consume the .npz files this writes.  Usage:

    python tests/golden/make_golden.py [--only c1,c2,...] [--skip-fullres]

Per config the fixture holds
  * the inputs as bytes (objects, lights, params, noise/skybox flags) -- so a test never depends
    on the scene generator staying bit-stable,
  * `lowres_*`: a complete low-resolution frame of the config's scene (all three surfaces),
  * `win_*`: eight 32x32 windows cut from the FULL-resolution frame of the config
    (GL compute has no dispatch offset, so the full frame is rendered and cropped),
  * `tan_bits`: llvmpipe's own value of tan(radians(fov)*0.5) (probe mode), which lets a test
    feed the restatement the same constant and compare the rest bit-for-bit.
Also writes probes.npz: llvmpipe outputs of the GLSL built-ins / expression shapes whose
lowering the restatement depends on.
"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)

from opengl_raytracing_amd import host, scenes  # noqa: E402
from oracle import binding as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
WIN = 32
WINDOW_FRACS = [(0.10, 0.10), (0.50, 0.20), (0.80, 0.15), (0.30, 0.45),
                (0.60, 0.50), (0.45, 0.30), (0.20, 0.70), (0.70, 0.75)]
# config -> (low-res size, render full resolution?)
PLAN = {
    "c1": (1, (128, 128), True),
    "c2": (2, (240, 135), True),
    "c3": (3, (240, 135), True),
    "c4": (4, (160, 90), True),
    "c5": (5, (160, 90), False),   # 8K x 256 objects x depth 8 is ~2 h on 8 vCPUs: low-res + 1080p only
    "nan": ("nan", (96, 96), False),
}


def probe_tan(fov_deg):
    glsl = """#version 430 core
layout(local_size_x = 1) in;
layout(std430, binding = 0) buffer In { float xin[]; };
layout(std430, binding = 1) buffer Out { float xout[]; };
void main() { xout[0] = tan(radians(xin[0]) * 0.5); }
"""
    out = O.run_probe(glsl, np.array([fov_deg], dtype=np.float32), np.float32, 1, 1)
    return int(out.view(np.int32)[0])


def windows(w, h):
    res = []
    for fx, fy in WINDOW_FRACS:
        x0 = min(max(int(fx * w) - WIN // 2, 0), w - WIN)
        y0 = min(max(int(fy * h) - WIN // 2, 0), h - WIN)
        res.append((x0, y0))
    return res


def params_bytes(p):
    return np.frombuffer(bytes(p), dtype=np.uint8).copy()


def run(sc, p, tag):
    t = time.time()
    col, pos, nrm, info = O.run_reference(sc, p, repeat=0)
    print(f"  [{tag}] {p.width}x{p.height} depth {p.maxRayDepth}: llvmpipe {info['first_dispatch_s']:.2f}s "
          f"(wall {time.time() - t:.1f}s)", flush=True)
    return col, pos, nrm.astype(np.float16), info


def make_config(name, skip_fullres):
    cfg, lowres, full = PLAN[name]
    sc = scenes.nan_parity_scene(host.generate_aabb) if cfg == "nan" else scenes.make_scene(cfg, host.generate_aabb)
    data = {
        "objects": np.frombuffer(sc.objects.tobytes(), dtype=np.uint8).copy(),
        "lights": np.frombuffer(sc.lights.tobytes(), dtype=np.uint8).copy(),
        "frame_count": np.int32(sc.frame_count),
        "has_noise": np.int32(sc.noise is not None),
        "has_skybox": np.int32(bool(sc.use_skybox)),
        "full_size": np.array([sc.width, sc.height], dtype=np.int32),
        "max_ray_depth": np.int32(sc.max_ray_depth),
    }
    p = sc.params(width=lowres[0], height=lowres[1])
    data["tan_bits"] = np.int32(probe_tan(p.fovDeg))
    col, pos, nrm, info = run(sc, p, name + " lowres")
    data["lowres_params"] = params_bytes(p)
    data["lowres_color"], data["lowres_pos"], data["lowres_normal"] = col, pos, nrm
    data["renderer"] = np.array(info["renderer"] + " / " + info["version"])
    if full and not skip_fullres:
        p = sc.params()
        col, pos, nrm, info = run(sc, p, name + " full")
        wins = windows(sc.width, sc.height)
        data["win_params"] = params_bytes(p)
        data["win_origins"] = np.array(wins, dtype=np.int32)
        data["win_color"] = np.stack([col[y:y + WIN, x:x + WIN] for x, y in wins])
        data["win_pos"] = np.stack([pos[y:y + WIN, x:x + WIN] for x, y in wins])
        data["win_normal"] = np.stack([nrm[y:y + WIN, x:x + WIN] for x, y in wins])
        data["full_dispatch_s"] = np.float64(info["first_dispatch_s"])
    elif cfg == 5 and not skip_fullres:
        # C5's scene at 1920x1080 (chain the evidence through a downscale)
        p = sc.params(width=1920, height=1080)
        col, pos, nrm, info = run(sc, p, name + " 1080p")
        wins = windows(1920, 1080)
        data["win_params"] = params_bytes(p)
        data["win_origins"] = np.array(wins, dtype=np.int32)
        data["win_color"] = np.stack([col[y:y + WIN, x:x + WIN] for x, y in wins])
        data["win_pos"] = np.stack([pos[y:y + WIN, x:x + WIN] for x, y in wins])
        data["win_normal"] = np.stack([nrm[y:y + WIN, x:x + WIN] for x, y in wins])
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **data)


PROBE_HEAD = """#version 430 core
layout(local_size_x = 64) in;
layout(std430, binding = 0) buffer In { vec4 xin[]; };
layout(std430, binding = 1) buffer Out { float xout[]; };
"""


def make_probes():
    """llvmpipe's outputs for the expression shapes whose lowering the oracle mirrors."""
    rng = np.random.default_rng(20250222)
    n = 1024
    out = {}
    # 1. scalar built-ins / shapes: rows of (a, b, c, d)
    x = rng.uniform(0, 1, (n, 4)).astype(np.float32)
    x[:, 1] = rng.uniform(-3, 3, n)
    glsl = PROBE_HEAD + """
float haltonSequence(int index, int base) {
    float result = 0.0; float f = 1.0 / base; int i = index;
    while(i > 0) { result += f * (i % base); i = i / base; f = f / base; }
    return result;
}
void main() {
    uint i = gl_GlobalInvocationID.x;
    float a = xin[i].x, b = xin[i].y, c = xin[i].z, d = xin[i].w;
    xout[i*8u+0u] = a + (1.0 - a) * c;               // F0 + (1-F0)*p     (:241)
    xout[i*8u+1u] = c * (1.0 - a) + a;               // NdotV*(1-k)+k     (:236)
    xout[i*8u+2u] = pow(a, 2.0) * (c * c - 1.0) + 1.0; // NDF inner       (:231)
    xout[i*8u+3u] = c * c / (3.14159265359 * pow(b, 2.0)); // PI*pow(x,2), x*x used once (:231)
    xout[i*8u+4u] = pow(b, 5.0);                     // negative base -> NaN
    xout[i*8u+5u] = haltonSequence(int(i), 2);
    xout[i*8u+6u] = haltonSequence(int(i), 3);
    xout[i*8u+7u] = 1.0 - a * (a * (1.0 - d * d));   // refract's k
}
"""
    out["scalar_in"] = x
    out["scalar_out"] = O.run_probe(glsl, x, np.float32, n * 8, n // 64).reshape(n, 8)
    # 2. vec3 built-ins: pairs of rows (a.xyz, t) (b.xyz, _)
    v = rng.uniform(-1, 1, (2 * n, 4)).astype(np.float32)
    v[0::2, 3] = rng.uniform(0, 1, n)
    for key, expr in [("mix_var", "mix(a, b, t)"), ("mix_const", "mix(vec3(0.04), b, t)"),
                      ("normalize", "normalize(a)"), ("reflect", "reflect(a, b)"),
                      ("refract", "refract(normalize(a), normalize(b), 0.3 + t)"), ("cross", "cross(a, b)")]:
        glsl = PROBE_HEAD + """
void main() {
    uint i = gl_GlobalInvocationID.x;
    vec3 a = xin[2u*i].xyz; float t = xin[2u*i].w; vec3 b = xin[2u*i+1u].xyz;
    vec3 m = %s;
    xout[i*4u+0u] = m.x; xout[i*4u+1u] = m.y; xout[i*4u+2u] = m.z; xout[i*4u+3u] = dot(a, b);
}
""" % expr
        out[key] = O.run_probe(glsl, v, np.float32, n * 4, n // 64).reshape(n, 4)
    out["vec_in"] = v
    np.savez_compressed(os.path.join(OUT, "probes.npz"), **out)
    print("  [probes] written", flush=True)


def make_taa():
    """TAA resolve fixture: the reference's taaFs.glsl + outputVs.glsl run on
    llvmpipe through gl_harness' postfx mode, fed with two consecutive oracle renders of the C2
    scene (the second with the objects moved, as history) and the first one's gNormal."""
    R = "/root/reference/shader/"
    sc = scenes.make_scene(2, host.generate_aabb)
    out = {}
    for tag, (W, H, fc, blend) in {"a": (96, 64, 5, 0.1), "b": (50, 37, 0, 0.5), "c": (64, 48, 3, 0.25)}.items():
        p = sc.params(width=W, height=H)
        cur, _, nrm, _ = O.render(sc, p)
        moved = scenes.make_scene(2, host.generate_aabb)
        moved.objects["position"][:, 0] += 0.07
        host.generate_aabb(moved.objects)
        hist, _, _, _ = O.render(moved, p)
        jx, jy = host.taa_jitter(fc, W, H)
        res = O.run_postfx(R + "outputVs.glsl", R + "taaFs.glsl", W, H,
                           [("uCurrentFrame", cur, dict(linear=True)), ("uHistory", hist, dict(linear=True, clamp=True)),
                            ("gNormal", nrm.astype(np.float32), dict(half=True))],
                           [("uBlendFactor", float(blend)), ("uJitterX", float(jx)), ("uJitterY", float(jy))])
        out[f"{tag}_current"], out[f"{tag}_history"], out[f"{tag}_normal"] = cur, hist, nrm
        out[f"{tag}_params"] = np.array([fc, blend, jx, jy], dtype=np.float64)
        out[f"{tag}_out"] = res
    np.savez_compressed(os.path.join(OUT, "taa.npz"), **out)
    print("  [taa] written", flush=True)


def make_bloom():
    """Bloom fixture: the reference's extract / blur / combine fragment shaders chained
    on llvmpipe exactly like ForwardShadingPipeline.cpp:189-228 (threshold 1.0, 10 alternating blur passes
    starting horizontal, strength 0.5), on an oracle render of the C2 scene scaled into HDR range."""
    R = "/root/reference/shader/"
    VS = R + "outputVs.glsl"
    sc = scenes.make_scene(2, host.generate_aabb)
    out = {}
    for tag, (W, H, gain) in {"a": (96, 64, 2.5), "b": (50, 37, 4.0)}.items():
        cur, _, _, _ = O.render(sc, sc.params(width=W, height=H))
        cur = (cur * np.float32(gain)).astype(np.float32)
        cur[..., 3] = 1.0
        tex = O.run_postfx(VS, R + "brightness_extractFS.glsl", W, H, [("hdrTexture", cur, dict(linear=True))],
                           [("threshold", 1.0)], out_half=True)
        stages = [tex]
        horizontal = True
        for _ in range(10):
            tex = O.run_postfx(VS, R + "gaussian_blurFs.glsl", W, H, [("image", tex, dict(half=True, linear=True, clamp=True))],
                               [("horizontal", int(horizontal))], out_half=True)
            stages.append(tex)
            horizontal = not horizontal
        comb = O.run_postfx(VS, R + "bloom_combineFs.glsl", W, H,
                            [("scene", cur, dict(linear=True)), ("bloomBlur", tex, dict(half=True, linear=True, clamp=True))],
                            [("bloomStrength", 0.5)])
        out[f"{tag}_scene"] = cur
        out[f"{tag}_stages"] = np.stack(stages).astype(np.float16)      # exact: the targets are rgba16f
        out[f"{tag}_combined"] = comb
    np.savez_compressed(os.path.join(OUT, "bloom.npz"), **out)
    print("  [bloom] written", flush=True)


def make_ssao():
    """SSAO fixture: ssaoFs.glsl and ssao_blurFs.glsl run unmodified on llvmpipe over oracle
    G-buffers (gPosition rgba32f, gNormal rgba16f, NEAREST / default REPEAT) with the glm view / projection
    matrices of Camera.h:36-42 and an AO.cpp:23-51-style kernel.  Also the llvmpipe lowering probes the
    restatement relies on (mat*vec association, projection*view*p = projection*(view*p), smoothstep).
    Image widths avoid W | 800*(x+0.5): there every pixel centre sits exactly on a rotation-texel boundary
    (TexCoords * 200 on a 4-texel REPEAT texture) and the reference's own result hangs on interpolation ulps."""
    R = "/root/reference/shader/"
    VS = R + "outputVs.glsl"
    samples, noise = host.ssao_kernel()
    out = {"samples": samples, "noise": noise}
    for tag, (cfg, W, H) in {"a": (2, 160, 90), "b": (5, 128, 64), "c": (1, 96, 54)}.items():
        sc = scenes.make_scene(cfg, host.generate_aabb)
        p = sc.params(width=W, height=H)
        _, pos, nrm, _ = O.render(sc, p)
        nrm = np.ascontiguousarray(nrm).view(np.float16).reshape(H, W, 4)
        if tag == "c":      # a patch of primary-miss texels: position 0, normal 0 (normalize -> NaN)
            pos[5:15, 10:30] = 0.0
            nrm[5:15, 10:30] = 0.0
        view, proj = host.camera_matrices(p.camPos[:], p.camDir[:], p.camUp[:], p.fovDeg, W / H)
        uni = [(f"samples[{i}]", tuple(samples[i])) for i in range(64)] + [("projection", tuple(proj)), ("view", tuple(view))]
        ao = O.run_postfx(VS, R + "ssaoFs.glsl", W, H, [("gPosition", pos, {}), ("gNormal", nrm.astype(np.float32), dict(half=True)),
                                                         ("texNoise", noise, {})], uni)[..., 0]
        rep = np.repeat(ao[..., None], 4, axis=2)
        blur_v = O.run_postfx(VS, R + "ssao_blurFs.glsl", W, H, [("ssaoInput", rep, {})], [("horizontal", 0)])[..., 0]
        blur_h = O.run_postfx(VS, R + "ssao_blurFs.glsl", W, H, [("ssaoInput", rep, {})], [("horizontal", 1)])[..., 0]
        out.update({f"{tag}_pos": pos, f"{tag}_nrm": nrm, f"{tag}_view": view, f"{tag}_proj": proj, f"{tag}_ao": ao,
                    f"{tag}_blur_v": blur_v, f"{tag}_blur_h": blur_h})
        print(f"  [ssao {tag}] cfg {cfg} {W}x{H}: mean ao {np.nanmean(ao):.4f}, NaN {np.isnan(ao).mean():.4f}", flush=True)
    # lowering probes (micro-kernels): M*v, A*B*w, smoothstep on random data
    n = 1024
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((n, 48)) * rng.choice([0.01, 1, 30], (n, 48))).astype(np.float32)
    glsl = """#version 430
layout(local_size_x=64) in;
layout(std430,binding=0) buffer I {float a[];};
layout(std430,binding=1) buffer Oo {float o[];};
void main(){ uint g=gl_GlobalInvocationID.x; uint b=g*48u;
 mat3 M=mat3(vec3(a[b],a[b+1],a[b+2]),vec3(a[b+3],a[b+4],a[b+5]),vec3(a[b+6],a[b+7],a[b+8]));
 vec3 v=vec3(a[b+9],a[b+10],a[b+11]);
 mat4 A,B; for(int c=0;c<4;c++){A[c]=vec4(a[b+12+c*4],a[b+13+c*4],a[b+14+c*4],a[b+15+c*4]);B[c]=vec4(a[b+28+c*4],a[b+29+c*4],a[b+30+c*4],a[b+31+c*4]);}
 vec4 w=vec4(a[b+44],a[b+45],a[b+46],1.0);
 vec3 r=M*v; vec4 q=A*B*w;
 float sm=smoothstep(0.0,1.0,0.5/abs(a[b+44]-a[b+45]));
 uint ob=g*8u; o[ob]=r.x;o[ob+1]=r.y;o[ob+2]=r.z; o[ob+3]=q.x;o[ob+4]=q.y;o[ob+5]=q.z;o[ob+6]=q.w; o[ob+7]=sm; }"""
    out["probe_in"] = x
    out["probe_out"] = O.run_probe(glsl, x, np.float32, n * 8, n // 64).reshape(n, 8)
    np.savez_compressed(os.path.join(OUT, "ssao.npz"), **out)
    print("  [ssao] written", flush=True)


def _look_at(eye, center, up):
    """glm::lookAtRH, column-major float32[16]."""
    f32 = np.float32
    eye, center, up = (np.asarray(a, f32) for a in (eye, center, up))
    nrm = lambda v: (v * (f32(1) / np.sqrt(np.dot(v, v).astype(f32)))).astype(f32)
    f = nrm(center - eye)
    s_ = nrm(np.cross(f, up).astype(f32))
    u = np.cross(s_, f).astype(f32)
    m = np.zeros((4, 4), f32)
    m[0][0], m[1][0], m[2][0] = s_
    m[0][1], m[1][1], m[2][1] = u
    m[0][2], m[1][2], m[2][2] = -f
    m[3][0], m[3][1], m[3][2], m[3][3] = -np.dot(s_, eye), -np.dot(u, eye), np.dot(f, eye), 1
    return m.ravel()


def _perspective(fovy, aspect, zn, zf):
    f32 = np.float32
    t = f32(np.tan(f32(fovy) / f32(2)))
    m = np.zeros((4, 4), f32)
    m[0][0], m[1][1] = f32(1) / (f32(aspect) * t), f32(1) / t
    m[2][2], m[2][3] = -(f32(zf) + f32(zn)) / (f32(zf) - f32(zn)), -1
    m[3][2] = -(f32(2) * f32(zf) * f32(zn)) / (f32(zf) - f32(zn))
    return m.ravel()


def make_cubemap():
    """Equirect -> cubemap fixture: skyboxVs.glsl + skyboxFs.glsl drawn over a unit cube with
    the six captureViews and the 90-degree captureProjection of TextureLoader.cpp:158-167, on a synthetic HDR
    panorama whose texels are fp16-representable (so the RGB16F upload is exact).  Face sizes are even: with an
    odd size the centre column of the -X face sits exactly on the atan seam (z = +-0) and the reference's own
    texels there flip between the panorama's two edges.  Plus probes of llvmpipe's atan(y,x) / asin lowering and
    of the float -> RGB16F upload rounding."""
    R = "/root/reference/shader/"
    f32 = np.float32
    W, H = 64, 32
    rng = np.random.default_rng(2)
    yy, xx = np.mgrid[0:H, 0:W]
    eqr = np.stack([1 + np.sin(xx / 5.0) + yy / 8.0, 2 + np.cos(yy / 3.0) * np.sin(xx / 7.0), (xx + yy) % 7 / 2.0], -1)
    eqr = ((eqr + rng.uniform(0, 0.3, (H, W, 3))) ** 2).astype(np.float16).astype(f32)
    tex = np.concatenate([eqr, np.ones((H, W, 1), f32)], -1)
    proj = _perspective(np.radians(f32(90.0)), 1.0, 0.1, 10.0)
    views = [((1, 0, 0), (0, -1, 0)), ((-1, 0, 0), (0, -1, 0)), ((0, 1, 0), (0, 0, 1)), ((0, -1, 0), (0, 0, -1)),
             ((0, 0, 1), (0, -1, 0)), ((0, 0, -1), (0, -1, 0))]
    out = {"equirect": eqr}
    for S in (32, 20, 128):
        faces = []
        for tg, up in views:
            ref = O.run_postfx(R + "skyboxVs.glsl", R + "skyboxFs.glsl", S, S,
                               [("equirectangularMap", tex, dict(half=True, linear=True, clamp=True))],
                               [("projection", tuple(proj)), ("view", tuple(_look_at((0, 0, 0), tg, up)))], out_half=True, cube=True)
            assert (ref[..., 3] == 1).all()
            faces.append(ref[..., :3].astype(np.float16))      # exact: the target is rgba16f
        out[f"faces_{S}"] = np.stack(faces)
        print(f"  [cubemap] S={S} done", flush=True)
    n = 4096
    v = rng.standard_normal((n, 3)).astype(f32)
    v[:48] = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1], [1, 1, 0], [1, 0, 1]] * 6, f32)
    v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(f32)
    glsl = """#version 430
layout(local_size_x=64) in;
layout(std430,binding=0) buffer I {float a[];};
layout(std430,binding=1) buffer Oo {float o[];};
void main(){ uint g=gl_GlobalInvocationID.x; vec3 v=vec3(a[g*3u],a[g*3u+1u],a[g*3u+2u]);
 o[g*4u]=atan(v.z,v.x); o[g*4u+1u]=asin(v.y); vec2 uv=vec2(atan(v.z,v.x),asin(v.y)); uv*=vec2(0.1591,0.3183); uv+=0.5; o[g*4u+2u]=uv.x; o[g*4u+3u]=uv.y; }"""
    out["probe_v"] = v
    out["probe_out"] = O.run_probe(glsl, v, np.float32, n * 4, n // 64).reshape(n, 4)
    # upload rounding: arbitrary floats into an RGBA16F texture, read back texel-exact through the extract shader
    # (threshold -1: brightness_extractFS.glsl passes the colour through)
    up = np.abs(rng.standard_normal((16, 32, 4))).astype(f32) * rng.choice([1e-3, 1.0, 50.0], (16, 32, 1)).astype(f32)
    up[..., 3] = 1
    out["upload_in"] = up
    out["upload_out"] = O.run_postfx(R + "outputVs.glsl", R + "brightness_extractFS.glsl", 32, 16,
                                     [("hdrTexture", up, dict(half=True, linear=True))], [("threshold", -1.0)])
    np.savez_compressed(os.path.join(OUT, "cubemap.npz"), **out)
    print("  [cubemap] written", flush=True)


def make_c5_8k(strip_groups_y=24, column_groups_x=4, win=48):
    """C5 at its REAL size, 7680x4320 (VERDICT r1 'parity unpinned at 8K'): GL compute has no dispatch offset,
    but the group count is free -- glDispatchCompute(240, GY, 1) on the full-size images runs exactly the
    invocations the full dispatch runs for the bottom 32*GY rows (gl_GlobalInvocationID, imageSize() and hence uv
    and random()'s arguments are those of the 8K frame), and glDispatchCompute(GX, 135, 1) the left 32*GX columns
    over the whole height (gid.y up to 4319: the large random() arguments).  The fixture keeps `win`-pixel
    windows cut from the two regions; the raw strips stay in /tmp (tens of MB)."""
    sc = scenes.make_scene(5, host.generate_aabb)
    p = sc.params()
    W, H = p.width, p.height
    assert (W, H) == (7680, 4320)
    data = {"objects": np.frombuffer(sc.objects.tobytes(), dtype=np.uint8).copy(),
            "lights": np.frombuffer(sc.lights.tobytes(), dtype=np.uint8).copy(),
            "params": params_bytes(p), "tan_bits": np.int32(probe_tan(p.fovDeg)),
            "frame_count": np.int32(sc.frame_count), "max_ray_depth": np.int32(sc.max_ray_depth)}
    origins, cols, poss, nrms = [], [], [], []
    regions = [("strip", (W // 32, strip_groups_y), (0, 0, W, 32 * strip_groups_y)),
               ("column", (column_groups_x, (H + 31) // 32), (0, 0, 32 * column_groups_x, H))]
    rng = np.random.default_rng(58)
    for tag, groups, crop in regions:
        t = time.time()
        col, pos, nrm, info = O.run_reference(sc, p, repeat=0, groups=groups, crop=crop)
        nrm = nrm.astype(np.float16)
        print(f"  [c5_8k {tag}] groups {groups}: llvmpipe {info['first_dispatch_s']:.1f}s (wall {time.time() - t:.1f}s)", flush=True)
        np.savez_compressed(f"/tmp/c5_8k_{tag}.npz", color=col, pos=pos, normal=nrm)
        data[f"{tag}_dispatch_s"] = np.float64(info["first_dispatch_s"])
        rw, rh = crop[2], crop[3]
        # windows on a jittered grid over the region; keep those with content first (hits), then fill with the rest
        nx, ny = max(1, rw // 640), max(1, rh // 360)
        cand = []
        for gy_ in range(ny):
            for gx_ in range(nx):
                x0 = int(min(rw - win, gx_ * rw / nx + rng.integers(0, max(1, rw // nx - win))))
                y0 = int(min(rh - win, gy_ * rh / ny + rng.integers(0, max(1, rh // ny - win))))
                hits = float((pos[y0:y0 + win, x0:x0 + win, 3] == 1).mean())          # all pixels store alpha 1
                var = float(np.nanstd(col[y0:y0 + win, x0:x0 + win, :3]))
                cand.append((var, x0, y0))
        cand.sort(reverse=True)
        for var, x0, y0 in cand[:14]:
            origins.append((crop[0] + x0, crop[1] + y0))
            cols.append(col[y0:y0 + win, x0:x0 + win])
            poss.append(pos[y0:y0 + win, x0:x0 + win])
            nrms.append(nrm[y0:y0 + win, x0:x0 + win])
    data["win8k_origins"] = np.array(origins, dtype=np.int32)
    data["win8k_color"], data["win8k_pos"], data["win8k_normal"] = np.stack(cols), np.stack(poss), np.stack(nrms)
    data["renderer"] = np.array(info["renderer"] + " / " + info["version"])
    np.savez_compressed(os.path.join(OUT, "c5_8k.npz"), **data)
    print(f"  [c5_8k] {len(origins)} windows of {win}x{win} written", flush=True)


def make_trig():
    """llvmpipe's sin / cos / tan / log2 / exp2 / pow(x,5) / exp, bitwise (trig.npz).  Pins oracle/rt_oracle.c's
    mesa_* restatements (gallivm's cephes-style sincos with fused multiply-adds, polynomial log2 / exp2): the
    camera's tan(radians(fov)*0.5), cosineWeightedHemisphere's cos/sin(phi), random()'s sin(large), SSS's exp."""
    rng = np.random.default_rng(7)
    n = 1 << 16
    q = n // 8
    x = np.concatenate([rng.uniform(-8, 8, 2 * q), rng.uniform(-1e6, 1e6, 2 * q),
                        (np.arange(q) % 4096 - 2048) * np.float32(np.pi / 2) + rng.normal(0, 1e-3, q),
                        np.radians(np.linspace(0.01, 179.99, q)) * 0.5, rng.uniform(0, 1.6, q),
                        10.0 ** rng.uniform(-30, 9.0, q)]).astype(np.float32)
    x[:6] = [0, -0.0, np.inf, -np.inf, np.nan, 1e9]
    glsl = """#version 430 core
layout(local_size_x = 64) in;
layout(std430, binding = 0) buffer In { float xin[]; };
layout(std430, binding = 1) buffer Out { float xout[]; };
void main(){ uint i=gl_GlobalInvocationID.x; float a=xin[i]; xout[3u*i]=sin(a); xout[3u*i+1u]=cos(a); xout[3u*i+2u]=tan(a); }
"""
    out = {"trig_in": x, "trig_out": O.run_probe(glsl, x, np.float32, 3 * n, n // 64).reshape(n, 3)}
    # the exact expression of generateCameraRay (:209) over fov in (0, 180)
    fov = np.concatenate([np.linspace(0.05, 179.95, n - 64), np.array([20, 30, 45, 60, 75, 90, 100, 120] * 8)]).astype(np.float32)
    glsl = """#version 430 core
layout(local_size_x = 64) in;
layout(std430, binding = 0) buffer In { float xin[]; };
layout(std430, binding = 1) buffer Out { float xout[]; };
void main(){ uint i=gl_GlobalInvocationID.x; xout[i]=tan(radians(xin[i]) * 0.5); }
"""
    out["fov_in"] = fov
    out["fov_tan"] = O.run_probe(glsl, fov, np.float32, n, n // 64)
    y = np.concatenate([rng.uniform(0, 1, n // 2), rng.uniform(-20, 20, n // 4), 10.0 ** rng.uniform(-20, 10, n // 4)]).astype(np.float32)
    y[:6] = [0, -0.0, np.inf, -np.inf, np.nan, 1.0]
    glsl = """#version 430 core
layout(local_size_x = 64) in;
layout(std430, binding = 0) buffer In { float xin[]; };
layout(std430, binding = 1) buffer Out { float xout[]; };
void main(){ uint i=gl_GlobalInvocationID.x; float a=xin[i]; xout[4u*i]=log2(a); xout[4u*i+1u]=exp2(a); xout[4u*i+2u]=pow(a,5.0); xout[4u*i+3u]=exp(a); }
"""
    out["explog_in"] = y
    out["explog_out"] = O.run_probe(glsl, y, np.float32, 4 * n, n // 64).reshape(n, 4)
    np.savez_compressed(os.path.join(OUT, "trig.npz"), **out)
    print("  [trig] written", flush=True)


def make_surface_probes():
    """rgba16f imageStore rounding + cubemap sampling through a render-mode job with a tiny
    custom shader is not needed: both are exercised by the c5/nan fixtures.  (Kept as a hook.)"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--skip-fullres", action="store_true")
    args = ap.parse_args()
    if not O.harness_available():
        sys.exit("gl_harness or /root/reference is not available: goldens can only be generated in the build container")
    names = [s for s in args.only.split(",") if s] or list(PLAN) + ["probes", "taa", "bloom", "ssao", "cubemap", "c5_8k", "trig"]
    for nme in names:
        print(f"== {nme}", flush=True)
        if nme == "probes":
            make_probes()
        elif nme == "taa":
            make_taa()
        elif nme == "bloom":
            make_bloom()
        elif nme == "ssao":
            make_ssao()
        elif nme == "cubemap":
            make_cubemap()
        elif nme == "c5_8k":
            make_c5_8k()
        elif nme == "trig":
            make_trig()
        else:
            make_config(nme, args.skip_fullres)


if __name__ == "__main__":
    main()
