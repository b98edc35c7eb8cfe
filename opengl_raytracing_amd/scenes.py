"""Deterministic synthetic scenes for the five BASELINE.json configs.

The reference ships no benchmark scenes of these shapes; these generators produce the
same bytes every run (splitmix64, seed ``0x5EED0000 + cfg``) in the reference's own
SSBO layouts (layout.py), so the oracle, the HIP path and the llvmpipe run of the
reference shader all consume identical inputs.  AABBs come from the library's
``rt_generate_aabb`` (the restatement of /root/reference/src/SceneIO.h:75-104).
"""
from dataclasses import dataclass, field

import numpy as np

from . import layout as L

_M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def uniform(self, lo=0.0, hi=1.0):
        u = (self.next() >> 40) / float(1 << 24)  # 24-bit mantissa: exact in fp32
        return float(np.float32(lo + (hi - lo) * u))


@dataclass
class Scene:
    name: str
    objects: np.ndarray
    lights: np.ndarray
    width: int
    height: int
    max_ray_depth: int
    camera: dict = field(default_factory=dict)
    frame_count: int = 0
    noise: np.ndarray = None      # (h, w) uint8 or None (= shipped behaviour: sample reads 0)
    skybox: np.ndarray = None     # (6, size, size, 3) float16 or None
    use_skybox: bool = False

    def params(self, width=None, height=None, window=None, strips=None, max_ray_depth=None):
        w = self.width if width is None else width
        h = self.height if height is None else height
        d = self.max_ray_depth if max_ray_depth is None else max_ray_depth
        return L.make_params(w, h, d, frame_count=self.frame_count, use_skybox=int(self.use_skybox),
                             window=window, strips=strips, **self.camera)


CAMERA = dict(cam_pos=(0.0, 2.0, 9.0), cam_dir=(0.0, 0.0, -1.0), cam_up=(0.0, 1.0, 0.0),
              cam_right=(1.0, 0.0, 0.0), fov_deg=45.0)


def _concat(parts):
    """np.concatenate repacks padded structured dtypes; keep the 176-byte stride."""
    out = np.zeros(sum(len(p) for p in parts), dtype=parts[0].dtype)
    k = 0
    for p in parts:
        out[k:k + len(p)] = p
        k += len(p)
    return out


def _spheres(rng, n, sss_index=None):
    o = L.default_objects(n)
    for i in range(n):
        o[i]["type"] = L.SPHERE
        o[i]["position"] = (rng.uniform(-9, 9), rng.uniform(0, 5), rng.uniform(-12, 2))
        o[i]["radius"] = rng.uniform(0.4, 1.4)
        kind = i % 3
        if kind == 0:    # metallic
            o[i]["mat_type"] = L.MATERIAL_METALLIC
            o[i]["albedo"] = tuple(rng.uniform(0.5, 0.95) for _ in range(3))
            o[i]["metallic"] = 1.0
            o[i]["roughness"] = rng.uniform(0.05, 0.5)
            o[i]["ior"] = 1.0
        elif kind == 1:  # dielectric
            o[i]["mat_type"] = L.MATERIAL_DIELECTRIC
            o[i]["albedo"] = tuple(rng.uniform(0.7, 1.0) for _ in range(3))
            o[i]["roughness"] = 0.05
            o[i]["ior"] = 1.5
            o[i]["transparency"] = 0.95
        else:            # plastic
            o[i]["mat_type"] = L.MATERIAL_PLASTIC
            o[i]["albedo"] = tuple(rng.uniform(0.1, 0.9) for _ in range(3))
            o[i]["roughness"] = rng.uniform(0.2, 1.0)
            o[i]["ior"] = 1.0
        # even index: the UI's value-initialised 0 (ImGUIManager.cpp:61); odd: diffuse branch
        ds = rng.uniform(0.3, 1.0)
        o[i]["diffuseStrength"] = 0.0 if i % 2 == 0 else ds
    if sss_index is not None and n > sss_index:
        o[sss_index]["subsurfaceScatter"] = 0.5
        o[sss_index]["subsurfaceColor"] = (1.0, 0.6, 0.5)
        o[sss_index]["scatterDistance"] = 0.8
    return o


def _planes(two=True):
    o = L.default_objects(2 if two else 1)
    o[0]["type"] = L.PLANE
    o[0]["position"] = (0.0, -1.0, -4.0)
    o[0]["normal"] = (0.0, 1.0, 0.0)
    o[0]["size"] = (40.0, 40.0)
    o[0]["albedo"] = (0.8, 0.8, 0.8)
    o[0]["roughness"] = 0.6
    o[0]["diffuseStrength"] = 0.0
    if two:
        o[1]["type"] = L.PLANE
        o[1]["position"] = (0.0, 9.0, -14.0)
        o[1]["normal"] = (0.0, 0.0, 1.0)
        o[1]["size"] = (40.0, 20.0)
        o[1]["albedo"] = (0.7, 0.75, 0.8)
        o[1]["roughness"] = 0.4
        o[1]["diffuseStrength"] = 0.5
    return o


def _lights3(shadow_type):
    l = L.default_lights(3)
    l[0]["type"] = L.POINT
    l[0]["position"] = (0.0, 6.0, -3.0)
    l[0]["intensity"] = 8.0
    l[1]["type"] = L.DIRECTIONAL
    l[1]["direction"] = (0.5, -1.0, -0.5)   # /root/reference/res/Scene/default.scene:5
    l[1]["intensity"] = 3.0
    l[2]["type"] = L.AREA
    l[2]["position"] = (3.0, 8.0, -6.0)
    l[2]["direction"] = (0.0, 1.0, 0.0)     # sign quirk: +y lights what is below
    l[2]["intensity"] = 40.0
    l["shadowType"] = shadow_type
    return l


def _ring_lights(n=8):
    l = L.default_lights(n)
    for i in range(n):
        a = 2.0 * np.pi * i / n
        l[i]["type"] = L.AREA
        l[i]["position"] = (np.float32(8.0 * np.cos(a)), 9.0, np.float32(-5.0 + 8.0 * np.sin(a)))
        l[i]["direction"] = (0.0, 1.0, 0.0)
        l[i]["color"] = (1.0, np.float32(0.85 + 0.15 * np.cos(a)), np.float32(0.85 + 0.15 * np.sin(a)))
        l[i]["intensity"] = 30.0
    l["shadowType"] = L.SHADOW_PCF
    return l


def hash_noise(w=1024, h=1024, seed=0xB10E):
    """R8 'blue-noise' stand-in (the reference's res/textures/blue_noise.png is absent,
    /root/reference/.MISSING_LARGE_BLOBS:8): a fixed integer hash per texel."""
    y, x = np.mgrid[0:h, 0:w].astype(np.uint64)
    v = (x * np.uint64(0x9E3779B1) + y * np.uint64(0x85EBCA77) + np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    v ^= v >> np.uint64(15)
    v = (v * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    v ^= v >> np.uint64(12)
    v = (v * np.uint64(0x297A2D39)) & np.uint64(0xFFFFFFFF)
    v ^= v >> np.uint64(15)
    return (v & np.uint64(0xFF)).astype(np.uint8)


def procedural_skybox(size=512):
    """6 x size^2 RGB fp16 faces in GL order (+X,-X,+Y,-Y,+Z,-Z), the format
    /root/reference/src/TextureLoader.cpp:140-147 allocates (the .hdr sources are
    absent, .MISSING_LARGE_BLOBS:1-7): a smooth sky gradient + sun lobe + bands."""
    faces = np.zeros((6, size, size, 3), dtype=np.float32)
    t = (np.arange(size, dtype=np.float32) + 0.5) / size * 2.0 - 1.0
    sc, tc = np.meshgrid(t, t)  # sc along x (s), tc along y (t)
    one = np.ones_like(sc)
    dirs = [
        (one, -tc, -sc), (-one, -tc, sc), (sc, one, tc), (sc, -one, -tc), (sc, -tc, one), (-sc, -tc, -one),
    ]
    sun = np.array([0.3, 0.6, -0.74], dtype=np.float32)
    sun /= np.linalg.norm(sun)
    for f, (dx, dy, dz) in enumerate(dirs):
        n = np.sqrt(dx * dx + dy * dy + dz * dz)
        dx, dy, dz = dx / n, dy / n, dz / n
        up = 0.5 * (dy + 1.0)
        base = np.stack([0.25 + 0.35 * up, 0.35 + 0.45 * up, 0.55 + 0.45 * up], axis=-1)
        lobe = np.clip(dx * sun[0] + dy * sun[1] + dz * sun[2], 0.0, 1.0) ** 64
        bands = 0.05 * np.sin(12.0 * np.arctan2(dz, dx))[..., None]
        faces[f] = base + bands + lobe[..., None] * np.array([6.0, 5.0, 3.5], dtype=np.float32)
    return faces.astype(np.float16)


def make_scene(cfg, generate_aabb):
    """cfg in 1..5 -> BASELINE.json configs[cfg-1].  ``generate_aabb(objects)`` fills
    Object.bounds in place (host.generate_aabb)."""
    rng = SplitMix64(0x5EED0000 + cfg)
    if cfg == 1:
        objs = _concat([_spheres(rng, 1), _planes(two=False)])
        objs[0]["position"] = (0.0, 1.0, -2.0)
        objs[0]["radius"] = 1.5
        lights = L.default_lights(1)
        lights[0]["type"] = L.POINT
        lights[0]["position"] = (2.0, 6.0, 2.0)
        lights[0]["intensity"] = 8.0
        sc = Scene("c1_1sphere_1plane_256", objs, lights, 256, 256, 1, dict(CAMERA))
    elif cfg == 2:
        objs = _concat([_spheres(rng, 16, sss_index=5), _planes()])
        sc = Scene("c2_16spheres_1080p", objs, _lights3(L.SHADOW_PCF), 1920, 1080, 4, dict(CAMERA))
    elif cfg == 3:
        # "same scene" as C2 (same seed stream), PCSS + noise texture + frameCount>0
        rng = SplitMix64(0x5EED0000 + 2)
        objs = _concat([_spheres(rng, 16, sss_index=5), _planes()])
        sc = Scene("c3_16spheres_pcss_4k", objs, _lights3(L.SHADOW_PCSS), 3840, 2160, 4, dict(CAMERA),
                   frame_count=7, noise=hash_noise())
    elif cfg == 4:
        objs = _concat([_spheres(rng, 62, sss_index=5), _planes()])
        sc = Scene("c4_64obj_8area_4k", objs, _ring_lights(8), 3840, 2160, 8, dict(CAMERA))
    elif cfg == 5:
        objs = _concat([_spheres(rng, 254, sss_index=5), _planes()])
        sc = Scene("c5_256obj_8k_sky", objs, _ring_lights(8), 7680, 4320, 8, dict(CAMERA),
                   skybox=procedural_skybox(512), use_skybox=True)
    else:
        raise ValueError("cfg must be 1..5")
    generate_aabb(sc.objects)
    return sc


def nan_parity_scene(generate_aabb):
    """Tiny scene that forces the reference's undefined-arithmetic corners
    (SURVEY.md A.1#12, a16, a17): roughness 0 (0/0 in the NDF), a light straight
    above (PCF tangent = normalize(0)), ior 0 refraction, tilted plane normal."""
    objs = L.default_objects(4)
    objs[0]["type"] = L.SPHERE
    objs[0]["position"] = (-1.5, 0.5, -4.0)
    objs[0]["radius"] = 1.0
    objs[0]["roughness"] = 0.0
    objs[0]["metallic"] = 1.0
    objs[1]["type"] = L.SPHERE
    objs[1]["position"] = (1.5, 0.5, -4.0)
    objs[1]["radius"] = 1.0
    objs[1]["ior"] = 0.0
    objs[1]["transparency"] = 0.95
    objs[1]["roughness"] = 0.05
    objs[2]["type"] = L.PLANE
    objs[2]["position"] = (0.0, -1.0, -4.0)
    objs[2]["normal"] = (0.0, 1.0, 0.0)
    objs[2]["size"] = (12.0, 12.0)
    objs[2]["diffuseStrength"] = 0.7
    objs[3]["type"] = L.PLANE
    objs[3]["position"] = (0.0, 2.0, -9.0)
    objs[3]["normal"] = (0.0, 0.0, 2.0)      # unnormalised normal (A.1#8)
    objs[3]["size"] = (12.0, 6.0)
    lights = L.default_lights(2)
    lights[0]["type"] = L.DIRECTIONAL
    lights[0]["direction"] = (0.0, -1.0, 0.0)  # lightDir || y -> NaN tangent (A.1#12)
    lights[0]["intensity"] = 2.0
    lights[1]["type"] = L.POINT
    lights[1]["position"] = (0.0, 4.0, -2.0)
    lights[1]["intensity"] = 6.0
    generate_aabb(objs)
    cam = dict(CAMERA)
    cam["cam_pos"] = (0.0, 1.0, 4.0)
    return Scene("nan_parity", objs, lights, 96, 96, 3, cam)
