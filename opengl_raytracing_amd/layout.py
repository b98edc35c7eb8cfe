"""Byte layouts of the reference's SSBO records and uniform block.

These mirror, field for field, the host structs the reference uploads with
``glBufferData`` and the std430 view its shader has of them:

* ``Object``  176 B  -- /root/reference/src/Object.h:13-21, shader/raytracingCs.glsl:34-42
* ``Material`` 80 B  -- /root/reference/src/Material.h:11-23 (embedded at byte 64)
* ``AABB``     32 B  -- /root/reference/src/Object.h:8-11   (embedded at byte 144)
* ``Light``    96 B  -- /root/reference/src/Light.h:7-20, shader/raytracingCs.glsl:44-58

Offsets are SURVEY.md Appendix B (host ``offsetof`` == llvmpipe GL_OFFSET).  The C
side (include/rt_mi355.h) ``static_assert``s the same numbers.
"""
import ctypes

import numpy as np

OBJECT_STRIDE = 176
LIGHT_STRIDE = 96

# ObjectType  (/root/reference/src/Object.h:6)
SPHERE, PLANE = 0, 1
# LightType   (/root/reference/src/Light.h:5)
POINT, DIRECTIONAL, AREA = 0, 1, 2
# shadowType  (/root/reference/src/Light.h:16)
SHADOW_NONE, SHADOW_PCF, SHADOW_PCSS = 0, 1, 2
# MaterialType (/root/reference/src/Material.h:5-9) -- never read by the shader
MATERIAL_METALLIC, MATERIAL_DIELECTRIC, MATERIAL_PLASTIC = 0, 1, 2

OBJECT_DTYPE = np.dtype(
    {
        "names": [
            "type", "position", "radius", "normal", "size",
            "mat_type", "albedo", "metallic", "roughness", "diffuseStrength", "ior",
            "transparency", "specular", "subsurfaceScatter", "subsurfaceColor",
            "scatterDistance", "bounds_min", "bounds_max",
        ],
        "formats": [
            "<i4", ("<f4", 3), "<f4", ("<f4", 3), ("<f4", 2),
            "<i4", ("<f4", 3), "<f4", "<f4", "<f4", "<f4",
            "<f4", "<f4", "<f4", ("<f4", 3),
            "<f4", ("<f4", 3), ("<f4", 3),
        ],
        "offsets": [0, 16, 28, 32, 48, 64, 80, 92, 96, 100, 104, 108, 112, 116, 128, 140, 144, 160],
        "itemsize": OBJECT_STRIDE,
    }
)

LIGHT_DTYPE = np.dtype(
    {
        "names": [
            "type", "position", "direction", "color", "intensity", "radius", "samples",
            "shadowSoftness", "shadowType", "pcfSamples", "lightSize", "angularRadius",
        ],
        "formats": [
            "<i4", ("<f4", 3), ("<f4", 3), ("<f4", 3), "<f4", "<f4", "<i4",
            "<f4", "<i4", "<i4", "<f4", "<f4",
        ],
        "offsets": [0, 16, 32, 48, 60, 64, 68, 72, 76, 80, 84, 88],
        "itemsize": LIGHT_STRIDE,
    }
)

assert OBJECT_DTYPE.itemsize == 176 and LIGHT_DTYPE.itemsize == 96


def default_objects(n):
    """n Objects carrying the reference's member initialisers
    (/root/reference/src/Object.h:16-18, Material.h:12-22); diffuseStrength, which
    the reference leaves uninitialised, is set to the UI's value-initialised 0."""
    o = np.zeros(n, dtype=OBJECT_DTYPE)
    o["radius"] = 1.0
    o["normal"] = (0.0, 1.0, 0.0)
    o["size"] = (1.0, 1.0)
    o["mat_type"] = MATERIAL_PLASTIC
    o["albedo"] = 1.0
    o["roughness"] = 0.5
    o["ior"] = 1.0
    o["specular"] = 0.5
    o["subsurfaceColor"] = 1.0
    o["scatterDistance"] = 0.1
    return o


def default_lights(n):
    """n Lights carrying /root/reference/src/Light.h:8-19's initialisers."""
    l = np.zeros(n, dtype=LIGHT_DTYPE)
    l["direction"] = (0.0, -1.0, 0.0)
    l["color"] = 1.0
    l["intensity"] = 1.0
    l["radius"] = 0.5
    l["samples"] = 4
    l["shadowSoftness"] = 1.0
    l["shadowType"] = SHADOW_PCF
    l["pcfSamples"] = 4
    l["lightSize"] = 1.0
    return l


class RtParams(ctypes.Structure):
    """``rt_params`` of include/rt_mi355.h (identical to ``orc_params`` of
    oracle/rt_oracle.h): the shader's uniforms (raytracingCs.glsl:72-89, uploaded at
    /root/reference/src/ForwardShadingPipeline.cpp:155-166) + image size + window."""

    _fields_ = [
        ("camPos", ctypes.c_float * 3),
        ("camDir", ctypes.c_float * 3),
        ("camUp", ctypes.c_float * 3),
        ("camRight", ctypes.c_float * 3),
        ("fovDeg", ctypes.c_float),
        ("focalLength", ctypes.c_float),
        ("maxRayDistance", ctypes.c_float),
        ("noiseScale", ctypes.c_float * 2),
        ("frameCount", ctypes.c_int32),
        ("useSkybox", ctypes.c_int32),
        ("maxRayDepth", ctypes.c_int32),
        ("width", ctypes.c_int32),
        ("height", ctypes.c_int32),
        ("x0", ctypes.c_int32),
        ("y0", ctypes.c_int32),
        ("regionW", ctypes.c_int32),
        ("regionH", ctypes.c_int32),
        ("stripRows", ctypes.c_int32),
        ("stripCount", ctypes.c_int32),
        ("stripIndex", ctypes.c_int32),
        ("reserved0", ctypes.c_int32),
        ("stripCycleRows", ctypes.c_int32),
        ("stripOffsetRows", ctypes.c_int32),
    ]


assert ctypes.sizeof(RtParams) == 128


def make_params(width, height, max_ray_depth, cam_pos=(0.0, 0.0, 0.0), cam_dir=(0.0, 0.0, -1.0),
                cam_up=(0.0, 1.0, 0.0), cam_right=(1.0, 0.0, 0.0), fov_deg=45.0, focal_length=1.0,
                max_ray_distance=114514.0, noise_scale=(1.0 / 1024.0, 1.0 / 1024.0), frame_count=0,
                use_skybox=0, window=None, strips=None):
    """Uniform defaults follow the reference: camera /root/reference/src/Camera.h:9-20,
    focalLength/maxRayDistance raytracingCs.glsl:80,85, noiseScale
    ForwardShadingPipeline.cpp:164.  ``window`` = (x0, y0, w, h) in local-row space,
    ``strips`` = (stripRows, stripCount, stripIndex)."""
    p = RtParams()
    p.camPos[:] = cam_pos
    p.camDir[:] = cam_dir
    p.camUp[:] = cam_up
    p.camRight[:] = cam_right
    p.fovDeg = fov_deg
    p.focalLength = focal_length
    p.maxRayDistance = max_ray_distance
    p.noiseScale[:] = noise_scale
    p.frameCount = frame_count
    p.useSkybox = int(use_skybox)
    p.maxRayDepth = max_ray_depth
    p.width, p.height = width, height
    if window is None:
        window = (0, 0, width, height)
    p.x0, p.y0, p.regionW, p.regionH = window
    if strips is None:
        strips = (1, 1, 0)
    p.stripRows, p.stripCount, p.stripIndex = strips
    return p


def copy_params(p, **updates):
    q = RtParams.from_buffer_copy(bytes(p))
    for k, v in updates.items():
        if isinstance(v, (tuple, list)):
            getattr(q, k)[:] = v
        else:
            setattr(q, k, v)
    return q
