// rt_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the per-pixel ray
// tracer.  What is computed follows /root/reference/shader/raytracingCs.glsl function by
// function (cited below); how it is computed is MI355X-first:
//
//   * one 8x8-pixel tile per 64-lane wavefront, 4 waves (16x16 px) per workgroup;
//   * the scene is "compiled" once per rt_set_scene into SoA-ish float4 records
//     (hot = AABB + shape, cold = material, lights with per-light invariants folded,
//     Halton tables) and staged ONCE PER WORKGROUP into LDS; the traversal loop reads it
//     with wave-uniform (broadcast) ds_read_b128, never from HBM;
//   * traversal carries a hit INDEX, the 64-byte material is fetched once per closest
//     hit (the GLSL copies 80 B on every improvement -- same result);
//   * shadow / PCSS-blocker rays use an any-hit traversal that leaves the loop as soon as
//     the whole wavefront is occluded (exactly equivalent to the reference's closest-hit
//     `t < lightDistance` test);
//   * AABB misses skip the shape test wave-wide (exec-mask branch, s_cbranch_execz);
//   * stores: one float4 per lane to gColor/gPosition (8 lanes = one 128-B line), one
//     8-byte packed half4 (round-toward-zero) to gNormal.
//
// Arithmetic is fp32 in the reference's evaluation order (no FMA contraction: built with
// -ffp-contract=off; IEEE divide/sqrt) so results are comparable bit-for-bit with the CPU
// restatement in oracle/ wherever no transcendental is involved.
#include "rt_device.h"
#include "rt_mesa_math.h"
#include "rt_fastmath.h"

#include <hip/hip_fp16.h>

#define WAVE 64
#define BLOCK_THREADS 256 // exhaustive kernel: four waves, 16x16-pixel workgroup tile
#define TILE 16

namespace {

struct v3 { float x, y, z; };

__device__ __forceinline__ v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 V3(float4 q) { return V3(q.x, q.y, q.z); }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 operator/(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
// the same through rt_fastmath.h's shared reciprocal (correctly rounded for every input; hot call sites)
__device__ __forceinline__ v3 div3(v3 a, float s) { v3 r; rtf::div3(a.x, a.y, a.z, s, r.x, r.y, r.z); return r; }
__device__ __forceinline__ v3 operator-(v3 a) { return V3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ v3 splat(float s) { return V3(s, s, s); }
// llvmpipe lowering: dot = (z*z + y*y) + x*x
__device__ __forceinline__ float dot(v3 a, v3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
__device__ __forceinline__ float length(v3 a) { return rtf::sqrt(dot(a, a)); }
__device__ __forceinline__ v3 normalize(v3 a) { return a * rtf::rcp_sqrt(dot(a, a)); }   // v * (1.0/sqrt(dot)), both steps correctly rounded
__device__ __forceinline__ v3 cross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// mix(): Mesa lowers the built-in context-dependently; the two shapes the shader uses were
// probed bitwise on llvmpipe (tests/test_oracle_units.py, fixture tests/golden/probes.npz):
//   all-variable operands (:562)  -> a + t*(b-a);   constant first operand (:240) -> a*(1-t) + b*t
__device__ __forceinline__ v3 mix_fast(v3 a, v3 b, float t) { return a + (b - a) * t; }
__device__ __forceinline__ v3 mix_strict(v3 a, v3 b, float t) { return a * (1.0f - t) + b * t; }
__device__ __forceinline__ v3 reflect(v3 I, v3 N) { return I - N * (2.0f * dot(N, I)); }
__device__ __forceinline__ v3 refract(v3 I, v3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * (eta * (1.0f - d * d));
    if (k < 0.0f) return V3(0.0f, 0.0f, 0.0f);
    return I * eta - N * (eta * d + rtf::sqrt(k));
}
__device__ __forceinline__ float fract(float x) { return x - floorf(x); }
// pow(x, 5.0) (raytracingCs.glsl:222, :241): NaN for x < 0 as on the reference's GL
// (exp2(5*log2 x)); exact-product form otherwise (<= 2 ulp; same three multiplies as the
// oracle, and ~20x cheaper than powf on the VALU).
__device__ __forceinline__ float pow5(float x) {
    float x2 = x * x;
    float r = (x2 * x2) * x;
    return x < 0.0f ? __int_as_float(0x7fc00000) : r;
}

// sin of random() (:274, one per Russian-roulette decision) and exp of the SSS term (:334): the reference GL's
// own polynomials (rt_mesa_math.h), so roulette decisions are the reference's, bit for bit.  fp32 + FMA only.
__device__ __forceinline__ float det_sinf(float x) { return rtm::sin_(x); }
__device__ __forceinline__ float det_expf(float x) { return rtm::exp_(x); }

constexpr float PI_F = 3.14159265359f;  // raytracingCs.glsl:6

struct Ray { v3 o, d; };

// LDS views of the compiled scene
struct SceneLds {
    const float4 *hot;     // nObj * RT_HOT_F4
    const float4 *mat;     // nObj * RT_MAT_F4
    const float4 *lgt;     // nLt  * RT_LGT_F4
    const float *halton2;  // RT_HALTON_N
    const float *halton3;  // RT_HALTON_N
    unsigned long long *stats = nullptr;   // diagnostic counters (instrumented build only)
    const float4 *global = nullptr;        // the same records in global memory (scalar-load path)
    float *park = nullptr;                 // per-thread LDS parking area, RT_PARK_FLOATS floats x BLOCK_THREADS (packet kernel)
    int tileX = 0, tileY = 0;              // this workgroup's tile (packet kernel; wave-uniform)
    int lgtF4Base = 0, haltonFloatBase = 0;   // offsets of the light / Halton sections (float4 / float units)
    // Compact staging (packet kernel, scenes too large to keep whole in LDS at full occupancy): LDS holds only
    // the two bounds float4 of every object (stride 2); shape / material records read per lane by hit index come
    // from the global copy.  hotStride = float4 stride of `hot`, matF4Base = material section in `global`.
    bool compact = false;
    bool boundsLds = true;      // packet kernel profile: the cull passes' per-lane AABB reads come from LDS (else from the global copy)
    bool wedge = false;         // packet kernel profile: convergent-packet cull of point / area light shadow rays
    int straight = 0;           // packet kernel profile: which candidate loops use the predicate-algebra tests (rt_packet.inc RT_PK_STRAIGHT)
    bool split = false;         // packet kernel profile: octant-split culling of sign-straddling packets
    bool keepAabb = true;       // packet kernel profile: chunk 0's AABB lives in the lane's VGPRs (else re-read from LDS per cull pass)
    int hotStride = RT_HOT_F4, matF4Base = 0;
    int pcfTabF4 = -1;          // float4 index in `global` of the directional lights' PCF ray tables, -1 = not usable (noise bound)
    const unsigned *stab = nullptr;        // shadow tables (rt_shadowtab.inc): per-light headers, then the cells
};

// haltonSequence (raytracingCs.glsl:278-288); used by the scene compiler and as the
// fallback for indices beyond the table.
__device__ float halton_eval(int index, int base) {
    float result = 0.0f;
    float f = 1.0f / (float)base;
    int i = index;
    while (i > 0) {
        result += f * (float)(i % base);
        i = i / base;
        f = f / (float)base;
    }
    return result;
}

// ---- traversal -------------------------------------------------------------------------
// intersectAABB (:91-103) with invDir hoisted per ray (same IEEE value as the per-object
// recomputation in the GLSL).
__device__ __forceinline__ bool aabb_test(const Ray &r, v3 inv, float4 h0, float4 h1, float maxDist) {
    float t0x = (h0.x - r.o.x) * inv.x, t1x = (h1.x - r.o.x) * inv.x;
    float t0y = (h0.y - r.o.y) * inv.y, t1y = (h1.y - r.o.y) * inv.y;
    float t0z = (h0.z - r.o.z) * inv.z, t1z = (h1.z - r.o.z) * inv.z;
    float tMin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
    float tMax = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
    return (tMax >= tMin) && (tMin < maxDist) && (tMax > 0.0f);
}

// intersectSphere (:105-118) / intersectPlane (:120-153) on compiled records.
// Returns true with t when the shape reports a hit (before the caller's t>0 && t<minT).
__device__ __forceinline__ bool shape_test(const Ray &r, float a, const float4 *h, int type, float &t) {
    if (type == 0) {
        float4 h1 = h[1], h2 = h[2];
        v3 oc = r.o - V3(h2);
        float b = 2.0f * dot(oc, r.d);
        float c = dot(oc, oc) - h1.w;               // h1.w = radius*radius
        float disc = b * b - 4.0f * a * c;
        if (disc < 0.0f) return false;
        t = (-b - rtf::sqrt(disc)) / (2.0f * a);
        return t > 0.0f;
    } else if (type == 1) {
        float4 h2 = h[2], h3 = h[3];
        v3 n = V3(h3);
        float denom = dot(n, r.d);
        if (fabsf(denom) > 1e-6f) {
            t = dot(V3(h2) - r.o, n) / denom;
            if (t < 0.0f) return false;
            float4 h4 = h[4], h5 = h[5];
            v3 lo = (r.o + r.d * t) - V3(h2);
            float x = dot(lo, V3(h4)), z = dot(lo, V3(h5));
            if (fabsf(x) > h3.w || fabsf(z) > h4.w) return false;   // size/2.0 precomputed
            return true;
        }
        return false;
    }
    return false;
}

// intersectObjects (:155-196), closest hit.  Returns the object index or -1; t = minT.
template <int COUNT>
__device__ __forceinline__ int trace_closest(const SceneLds &sc, int nObj, const Ray &r, float maxDist,
                                              float &tOut, unsigned &rays) {
    if (COUNT) rays++;
    v3 inv;
    rtf::rcp3(r.d.x, r.d.y, r.d.z, inv.x, inv.y, inv.z);
    float a = dot(r.d, r.d);
    float minT = maxDist;
    int hit = -1;
    for (int i = 0; i < nObj; i++) {
        const float4 *h = sc.hot + i * RT_HOT_F4;
        float4 h0 = h[0], h1 = h[1];
        if (aabb_test(r, inv, h0, h1, maxDist)) {
            float t;
            if (shape_test(r, a, h, __float_as_int(h0.w), t) && t > 0.0f && t < minT) {
                minT = t;
                hit = i;
            }
        }
    }
    tOut = minT;
    return hit;
}

// Any-hit form for shadow / PCSS blocker rays: true iff some object is hit with
// 0 < t < limit, limit = min(maxRayDistance, lightDistance) for point/area lights and
// maxRayDistance for directional ones (exactly the reference's closest-hit test, A.1#14).
// Lanes that found an occluder idle; the loop ends when the whole wave is done.
template <int COUNT>
__device__ __forceinline__ bool trace_any(const SceneLds &sc, int nObj, const Ray &r, float maxDist,
                                           float limit, unsigned &rays) {
    if (COUNT) rays++;
    v3 inv;
    rtf::rcp3(r.d.x, r.d.y, r.d.z, inv.x, inv.y, inv.z);
    float a = dot(r.d, r.d);
    bool occ = false;
    for (int i = 0; i < nObj; i++) {
        const float4 *h = sc.hot + i * RT_HOT_F4;
        float4 h0 = h[0], h1 = h[1];
        if (!occ && aabb_test(r, inv, h0, h1, maxDist)) {
            float t;
            if (shape_test(r, a, h, __float_as_int(h0.w), t) && t > 0.0f && t < limit) occ = true;
        }
        if (__builtin_amdgcn_ballot_w64(!occ) == 0ull) break;   // whole wavefront occluded
    }
    return occ;
}

// ---- textures --------------------------------------------------------------------------
// texture(blueNoiseTex, (gid + frameCount) * noiseScale).r : R8, NEAREST, REPEAT (:513, :359)
__device__ __forceinline__ float sample_noise(const RtFrame &f, const uint8_t *noise, unsigned gx, unsigned gy) {
    if (!noise) return 0.0f;
    float u = (float)(gx + (unsigned)f.p.frameCount) * f.p.noiseScale[0];
    float v = (float)(gy + (unsigned)f.p.frameCount) * f.p.noiseScale[1];
    u = fract(u); v = fract(v);
    int ix = (int)floorf(u * (float)f.noiseW), iy = (int)floorf(v * (float)f.noiseH);
    ix = min(max(ix, 0), f.noiseW - 1);
    iy = min(max(iy, 0), f.noiseH - 1);
    return (float)noise[(size_t)iy * f.noiseW + ix] * (1.0f / 255.0f);
}

__device__ __forceinline__ float half_bits_to_float(uint16_t h) {
    return __half2float(__ushort_as_half(h));
}

// texture(skybox, d).rgb : LINEAR, CLAMP_TO_EDGE per face, non-seamless, RGB16F.  Manual fp32
// bilinear (hardware filtering's 8-bit weights would break the 1e-4 tolerance; A.3).
__device__ v3 sample_cube(const uint16_t *sky, int size, v3 d) {
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int face; float sc, tc, ma;
    if (ax >= ay && ax >= az) {
        ma = ax;
        if (d.x >= 0.0f) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; }
    } else if (ay >= az) {
        ma = ay;
        if (d.y >= 0.0f) { face = 2; sc = d.x; tc = d.z; } else { face = 3; sc = d.x; tc = -d.z; }
    } else {
        ma = az;
        if (d.z >= 0.0f) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; }
    }
    float s = (sc / ma + 1.0f) / 2.0f, t = (tc / ma + 1.0f) / 2.0f;
    float u = s * (float)size - 0.5f, v = t * (float)size - 0.5f;
    float fu = floorf(u), fv = floorf(v);
    float wu = u - fu, wv = v - fv;
    int x0 = (int)fu, y0 = (int)fv;
    int x1 = min(max(x0 + 1, 0), size - 1), y1 = min(max(y0 + 1, 0), size - 1);
    x0 = min(max(x0, 0), size - 1); y0 = min(max(y0, 0), size - 1);
    const uint16_t *f = sky + (size_t)face * size * size * 3;
    const uint16_t *p00 = f + ((size_t)y0 * size + x0) * 3, *p10 = f + ((size_t)y0 * size + x1) * 3;
    const uint16_t *p01 = f + ((size_t)y1 * size + x0) * 3, *p11 = f + ((size_t)y1 * size + x1) * 3;
    float out[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float c00 = half_bits_to_float(p00[ch]), c10 = half_bits_to_float(p10[ch]);
        float c01 = half_bits_to_float(p01[ch]), c11 = half_bits_to_float(p11[ch]);
        float a = c00 + wu * (c10 - c00);
        float b = c01 + wu * (c11 - c01);
        out[ch] = a + wv * (b - a);
    }
    return V3(out[0], out[1], out[2]);
}

// imageStore to rgba16f rounds toward zero on the reference's GL (A.3).
__device__ __forceinline__ unsigned f2h_rtz(float f) {
    unsigned u = __float_as_uint(f);
    unsigned s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (a == 0x7f800000u) ? (s | 0x7c00u) : (s | 0x7e00u | ((a >> 13) & 0x1ffu));
    if (a >= 0x47800000u) return s | 0x7bffu;
    if (a >= 0x38800000u) return s | ((a - 0x38000000u) >> 13);
    if (a < 0x33800000u) return s;
    unsigned e = a >> 23, m = (a & 0x7fffffu) | 0x800000u;
    return s | (m >> (126u - e));
}

// ---- shading ---------------------------------------------------------------------------
struct Mat {
    v3 albedo; float metallic, roughness, diffuseStrength, ior, transparency;
    v3 sssColor; float sss, scatterDistance;
};

__device__ __forceinline__ Mat load_mat(const SceneLds &sc, int idx) {
    const float4 *m = sc.mat + idx * RT_MAT_F4;
    float4 m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3];
    Mat r;
    r.albedo = V3(m0); r.metallic = m0.w;
    r.roughness = m1.x; r.diffuseStrength = m1.y; r.ior = m1.z; r.transparency = m1.w;
    r.sssColor = V3(m2); r.sss = m2.w; r.scatterDistance = m3.x;
    return r;
}

// fresnelSchlick (:220-223); pow(x,2.0) folds to x*x on the reference's GL (A.3)
__device__ __forceinline__ float fresnel_schlick(float cosTheta, float ior) {
    float q = (1.0f - ior) / (1.0f + ior);
    float r0 = q * q;
    return r0 + (1.0f - r0) * pow5(1.0f - cosTheta);
}

// computePBR (:226-253).  Its nine divisions (NDF, the two G factors, vec3 / float twice) are 99 of its ~190 VALU
// instructions as IEEE sequences; FAST takes them through rt_fastmath.h's shared-reciprocal quotients (correctly rounded
// while `ok`, which is OR-ed over all nine) and the caller redoes the call with the IEEE divisions when a wave has a lane
// whose operands leave the fast range (a wave-uniform branch that C2..C5 never take: tools/gpu_fastmath_stats.py).
constexpr float RCP_PI_F = 0x1.45f306p-2f;      // RN(1 / PI_F), PI_F being the float the shader's 3.14159265359 rounds to
template <bool FAST>
__device__ __forceinline__ v3 compute_pbr_t(const Mat &m, v3 N, v3 V, v3 L, v3 H, v3 radiance, bool &ok) {
    float alpha = m.roughness * m.roughness;
    float NdotH = fmaxf(dot(N, H), 0.0f);
    // Mesa's NIR rewrites x*(a2-1)+1 as lerp(1, a2, x) = (1-x) + a2*x and PI*(i*i) as (PI*i)*i;
    // both probed bitwise on llvmpipe.  The first matters: at grazing highlights the as-written
    // form differs by up to 4e-4 rel in NDF.
    float nh2 = NdotH * NdotH;
    float inner = (1.0f - nh2) + (alpha * alpha) * nh2;
    const float ndfDen = (PI_F * inner) * inner;
    float rp1 = m.roughness + 1.0f;
    float k = (rp1 * rp1) / 8.0f;
    float NdotV = fmaxf(dot(N, V), 0.0f), NdotL = fmaxf(dot(N, L), 0.0f);
    const float gvDen = NdotV * (1.0f - k) + k, glDen = NdotL * (1.0f - k) + k;
    float NDF, G;
    ok = true;
    if constexpr (FAST) {
        bool o0, o1, o2, q0, q1, q2;
        const float y0 = rtf::rcp_fast(ndfDen, o0), y1 = rtf::rcp_fast(gvDen, o1), y2 = rtf::rcp_fast(glDen, o2);
        NDF = rtf::div_fast(alpha * alpha, ndfDen, y0, q0);
        G = rtf::div_fast(NdotV, gvDen, y1, q1);
        G *= rtf::div_fast(NdotL, glDen, y2, q2);
        ok = o0 & o1 & o2 & q0 & q1 & q2;
    } else {
        NDF = alpha * alpha / ndfDen;
        G = NdotV / gvDen;
        G *= NdotL / glDen;
    }
    v3 F0 = mix_strict(splat(0.04f), m.albedo, m.metallic);
    float p5 = pow5(1.0f - fmaxf(dot(H, V), 0.0f));
    v3 F = F0 + (splat(1.0f) - F0) * p5;
    v3 numerator = F * (NDF * G);
    float denominator = 4.0f * NdotV * NdotL;
    const float specDen = fmaxf(denominator, 0.001f);
    v3 kD = (splat(1.0f) - F) * (1.0f - m.metallic);
    const v3 dnum = kD * m.albedo;
    v3 specular, diffuse;
    if constexpr (FAST) {
        bool o3, s0, s1, s2, d0, d1, d2;
        const float y3 = rtf::rcp_fast(specDen, o3);
        specular = V3(rtf::div_fast(numerator.x, specDen, y3, s0), rtf::div_fast(numerator.y, specDen, y3, s1), rtf::div_fast(numerator.z, specDen, y3, s2));
        diffuse = V3(rtf::div_fast(dnum.x, PI_F, RCP_PI_F, d0), rtf::div_fast(dnum.y, PI_F, RCP_PI_F, d1), rtf::div_fast(dnum.z, PI_F, RCP_PI_F, d2));
        ok = ok & o3 & s0 & s1 & s2 & d0 & d1 & d2;
    } else {
        specular = numerator / specDen;
        diffuse = dnum / PI_F;
    }
    return ((diffuse + specular) * radiance) * NdotL;
}

#ifndef RT_FAST_DIV
#define RT_FAST_DIV 1           // 0: every general division of the path by the compiler's IEEE sequence
#endif
// `live`: lanes whose result is used (the packet kernel calls this with the whole wave; a dead lane's stale or NaN operands
// must not send the wave through the IEEE instantiation).  FASTDIV is a profile constant of the packet kernel (rt_packet.inc:
// RT_LIGHT_FAST_PBR / RT_HEAVY_FAST_PBR carry the measurements); every shipped profile uses it.
template <bool FASTDIV = false>
__device__ __forceinline__ v3 compute_pbr(const Mat &m, v3 N, v3 V, v3 L, v3 H, v3 radiance, bool live = true) {
    bool ok;
    if constexpr (FASTDIV && RT_FAST_DIV) {
        v3 r = compute_pbr_t<true>(m, N, V, L, H, radiance, ok);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(live && !ok) != 0ull, 0)) r = compute_pbr_t<false>(m, N, V, L, H, radiance, ok);
        return r;
    } else {
        return compute_pbr_t<false>(m, N, V, L, H, radiance, ok);
    }
}

// cosineWeightedHemisphere (:291-308) with the per-depth local direction h precomputed on
// the host (it depends only on depth and frameCount).
__device__ __forceinline__ v3 hemisphere_dir(v3 h, v3 n) {
    v3 tangent = normalize(cross(n, V3(0.0f, 1.0f, 1.0f)));
    v3 bitangent = cross(n, tangent);
    // tangent = (n.y - n.z, -n.x, n.x) * rsq: the reference's GL factors bitangent.x = n.y*t.z + n.z*t.z into t.z * (n.y + n.z)
    // (its final NIR, read with LP_DEBUG=cs; oracle/rt_oracle.c cosineWeightedHemisphere) -- C3's residue of round 2
    bitangent.x = tangent.z * (n.y + n.z);
    return normalize((tangent * h.x + bitangent * h.z) + n * h.y);
}

__device__ __forceinline__ float halton_lookup(const float *table, int i, int base) {
    return (i < RT_HALTON_N) ? table[i] : halton_eval(i, base);
}

// pcfShadow (:342-397)
template <int COUNT>
__device__ __forceinline__ float pcf_shadow(const SceneLds &sc, const RtFrame &f, v3 origin, int ltype,
                                             int pcfSamples, float filterSize, v3 lightDir, float limit,
                                             float jitterR, unsigned &rays) {
    float shadow = 0.0f;
    v3 tangent = normalize(cross(lightDir, V3(0.0f, 1.0f, 0.0f)));
    v3 bitangent = cross(lightDir, tangent);
    for (int i = 0; i < pcfSamples; i++) {
        float rx = fract(halton_lookup(sc.halton2, i, 2) + jitterR);
        float ry = fract(halton_lookup(sc.halton3, i, 3) + 0.0f);
        v3 jd = (lightDir + (tangent * rx) * filterSize) + (bitangent * ry) * filterSize;
        if (ltype != 1) jd = normalize(jd);
        Ray sr; sr.o = origin; sr.d = jd;
        bool occ = trace_any<COUNT>(sc, f.nObj, sr, f.p.maxRayDistance, limit, rays);
        shadow += occ ? 0.0f : 1.0f;
    }
    return shadow / (float)pcfSamples;
}

// pcssShadow (:400-440): 16 blocker-search rays; any blocker -> plain PCF (penumbra size is dead)
template <int COUNT>
__device__ __forceinline__ float pcss_shadow(const SceneLds &sc, const RtFrame &f, v3 origin, int ltype,
                                              int pcfSamples, float filterSize, float searchSize, v3 lightDir,
                                              float limit, float jitterR, unsigned &rays) {
    bool any = false;
    for (int i = 0; i < 16; i++) {
        float rr = sc.halton3[i] * 2.0f - 1.0f;
        v3 sd = (lightDir + splat(rr * searchSize)) + splat(rr * searchSize);
        Ray sr; sr.o = origin; sr.d = normalize(sd);
        any |= trace_any<COUNT>(sc, f.nObj, sr, f.p.maxRayDistance, limit, rays);
    }
    if (!any) return 1.0f;
    return pcf_shadow<COUNT>(sc, f, origin, ltype, pcfSamples, filterSize, lightDir, limit, jitterR, rays);
}

// computeSubsurfaceScattering (:316-339)
template <int COUNT>
__device__ v3 compute_sss(const SceneLds &sc, const RtFrame &f, v3 P, v3 N, const Mat &m, unsigned &rays) {
    v3 sss = V3(0.0f, 0.0f, 0.0f);
    for (int i = 0; i < 4; i++) {
        Ray r;
        r.o = P + N * 0.001f;
        r.d = hemisphere_dir(V3(f.sssHemi[i][0], f.sssHemi[i][1], f.sssHemi[i][2]), N);
        float t;
        int idx = trace_closest<COUNT>(sc, f.nObj, r, f.p.maxRayDistance, t, rays);
        if (idx >= 0) {
            float att = det_expf(-t / m.scatterDistance);
            sss = sss + V3(sc.mat[idx * RT_MAT_F4]) * att;
        }
    }
    return ((sss * m.sssColor) * m.sss) / 4.0f;
}

// computeLighting (:457-507)
template <int COUNT>
__device__ __forceinline__ v3 compute_lighting(const SceneLds &sc, const RtFrame &f, v3 P, v3 N, const Mat &m,
                                                v3 V, float jitterR, unsigned &rays) {
    v3 Lo = V3(0.0f, 0.0f, 0.0f);
    v3 shadowOrigin = P + N * 0.001f;     // :381, :412
    for (int i = 0; i < f.nLt; i++) {
        const float4 *lr = sc.lgt + i * RT_LGT_F4;
        float4 l0 = lr[0], l1 = lr[1], l2 = lr[2], l3 = lr[3];
        int ltype = __float_as_int(l0.w);
        v3 lightDir = V3(0.0f, 0.0f, 0.0f);
        float attenuation = 1.0f, lightDistance = 0.0f;
        if (ltype == 0) {
            lightDir = V3(l0) - P;
            lightDistance = length(lightDir);
            attenuation = rtf::rcp(1.0f + 0.1f * lightDistance + 0.01f * lightDistance * lightDistance);
            lightDir = normalize(lightDir);
        } else if (ltype == 1) {
            lightDir = V3(l1);             // normalize(-direction), folded by the scene compiler
            lightDistance = 1e6f;
        } else if (ltype == 2) {
            lightDir = V3(l0) - P;
            // lightDistance*lightDistance = sqrt(q)*sqrt(q) folds to |q| on the reference's GL
            float q = dot(lightDir, lightDir);
            lightDistance = length(lightDir);
            lightDir = normalize(lightDir);
            attenuation = rtf::rcp(fabsf(q));
            float lc = fmaxf(dot(lightDir, V3(l1)), 0.0f);   // l1 = normalize(direction)
            attenuation *= lc;
        }
        int shadowType = __float_as_int(l3.x), pcfSamples = __float_as_int(l3.y);
        float shadowFactor = 1.0f;
        if (shadowType != 0) {
            // any-hit limit: closest t < lightDistance for point/area lights (:390-392, :421-423)
            float limit = (ltype == 1) ? f.p.maxRayDistance : fminf(f.p.maxRayDistance, lightDistance);
            // NaN lightDistance: the reference's `t < NaN` is false -> never occluded
            if (ltype != 1 && !(lightDistance == lightDistance)) limit = -1.0f;
            if (shadowType == 1)
                shadowFactor = pcf_shadow<COUNT>(sc, f, shadowOrigin, ltype, pcfSamples, l2.w, lightDir, limit, jitterR, rays);
            else if (shadowType == 2)
                shadowFactor = pcss_shadow<COUNT>(sc, f, shadowOrigin, ltype, pcfSamples, l2.w, l3.z, lightDir, limit, jitterR, rays);
            else
                shadowFactor = 0.0f;       // calculateShadow's `float shadow = 0.0` fall-through (:445)
        }
        v3 L = normalize(lightDir);
        v3 H = normalize(V + L);
        v3 radiance = (V3(l2) * attenuation) * l1.w;
        Lo = Lo + compute_pbr(m, N, V, L, H, radiance) * shadowFactor;
    }
    if (m.sss > 0.0f) Lo = Lo + compute_sss<COUNT>(sc, f, P, N, m, rays);
    return Lo;
}

// calculateRefraction (:256-270)
__device__ __forceinline__ v3 calc_refraction(const Ray &r, v3 N, float ior) {
    bool entering = dot(r.d, N) < 0.0f;
    float eta = entering ? rtf::rcp(ior) : ior;
    v3 normal = entering ? N : -N;
    v3 rd = refract(normalize(r.d), normal, eta);
    if (dot(rd, rd) < 0.001f) rd = reflect(r.d, normal);
    return rd;
}

// random (:273-275)
__device__ __forceinline__ float random2(float sx, float sy) {
    float d = sy * 78.233f + sx * 12.9898f;
    return fract(det_sinf(d) * 43758.5453123f);
}

#include "rt_shadowtab.inc"
#include "rt_packet.inc"

}  // namespace

// =========================================================================================
// Scene compiler: raw std430 bytes -> compiled float4 records (one launch per rt_set_scene).
// =========================================================================================
__global__ void rt_compile_scene_kernel(const uint8_t *objects, int nObj, const uint8_t *lights, int nLt,
                                        float4 *out) {
    float4 *hot = out;
    float4 *mat = hot + (size_t)nObj * RT_HOT_F4;
    float4 *lgt = mat + (size_t)nObj * RT_MAT_F4;
    float *h2 = (float *)(lgt + (size_t)nLt * RT_LGT_F4);
    float *h3 = h2 + RT_HALTON_N;
    for (int i = threadIdx.x; i < nObj; i += blockDim.x) {
        const float *o = (const float *)(objects + (size_t)i * RT_OBJECT_STRIDE);   // offsets: Appendix B
        int type = ((const int *)o)[0];
        v3 pos = V3(o[4], o[5], o[6]);
        float radius = o[7];
        v3 n = V3(o[8], o[9], o[10]);
        float sx = o[12], sy = o[13];
        // plane basis of intersectPlane (:129-138): constant per object
        v3 right;
        if (fabsf(n.y) > 0.9f) right = normalize(cross(n, V3(0.0f, 0.0f, 1.0f)));
        else right = normalize(cross(n, V3(0.0f, 1.0f, 0.0f)));
        v3 forward = normalize(cross(right, n));
        float4 *h = hot + (size_t)i * RT_HOT_F4;
        h[0] = make_float4(o[36], o[37], o[38], __int_as_float(type));   // bounds.min @144
        h[1] = make_float4(o[40], o[41], o[42], radius * radius);         // bounds.max @160
        h[2] = make_float4(pos.x, pos.y, pos.z, radius);
        h[3] = make_float4(n.x, n.y, n.z, sx / 2.0f);
        h[4] = make_float4(right.x, right.y, right.z, sy / 2.0f);
        h[5] = make_float4(forward.x, forward.y, forward.z, 0.0f);
        float4 *m = mat + (size_t)i * RT_MAT_F4;
        m[0] = make_float4(o[20], o[21], o[22], o[23]);   // albedo @80, metallic @92
        m[1] = make_float4(o[24], o[25], o[26], o[27]);   // roughness, diffuseStrength, ior, transparency
        m[2] = make_float4(o[32], o[33], o[34], o[29]);   // subsurfaceColor @128, subsurfaceScatter @116
        m[3] = make_float4(o[35], 0.0f, 0.0f, 0.0f);      // scatterDistance @140
    }
    for (int i = threadIdx.x; i < nLt; i += blockDim.x) {
        const float *l = (const float *)(lights + (size_t)i * RT_LIGHT_STRIDE);
        int type = ((const int *)l)[0];
        v3 dir = V3(l[8], l[9], l[10]);
        v3 dn = (type == 1) ? normalize(-dir) : normalize(dir);   // :475 / :486
        float4 *r = lgt + (size_t)i * RT_LGT_F4;
        r[0] = make_float4(l[4], l[5], l[6], __int_as_float(type));
        r[1] = make_float4(dn.x, dn.y, dn.z, l[15]);                 // intensity @60
        r[2] = make_float4(l[12], l[13], l[14], l[18] * 0.005f);     // color @48, filterSize (:364)
        r[3] = make_float4(__int_as_float(((const int *)l)[19]),     // shadowType @76
                           __int_as_float(((const int *)l)[20]),     // pcfSamples @80
                           l[21] * 0.1f, 0.0f);                      // searchSize (:404)
    }
    for (int i = threadIdx.x; i < RT_HALTON_N; i += blockDim.x) {
        h2[i] = halton_eval(i, 2);
        h3[i] = halton_eval(i, 3);
    }
    // pcfShadow's jittered rays toward a DIRECTIONAL light (:352-375) do not depend on the shading point: with no
    // noise texture bound (noise.rg = 0) sample s of every pixel, bounce and frame is the same ray.  Tabulate
    // direction, dot(d,d) and 1/direction per light and sample with the expressions the per-lane path uses.
    float4 *tab = (float4 *)(h3 + RT_HALTON_N);
    for (int k = threadIdx.x; k < nLt * RT_PCF_TAB_N; k += blockDim.x) {
        const int li = k / RT_PCF_TAB_N, sidx = k % RT_PCF_TAB_N;
        const float *l = (const float *)(lights + (size_t)li * RT_LIGHT_STRIDE);
        const int type = ((const int *)l)[0];
        float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
        if (type == 1) {
            const v3 lightDir = normalize(-V3(l[8], l[9], l[10]));
            const float filterSize = l[18] * 0.005f;
            const v3 tangent = normalize(cross(lightDir, V3(0.0f, 1.0f, 0.0f)));
            const v3 bitangent = cross(lightDir, tangent);
            const float rx = fract(halton_eval(sidx, 2) + 0.0f), ry = fract(halton_eval(sidx, 3) + 0.0f);
            const v3 jd = (lightDir + (tangent * rx) * filterSize) + (bitangent * ry) * filterSize;
            t0 = make_float4(jd.x, jd.y, jd.z, dot(jd, jd));
            t1 = make_float4(1.0f / jd.x, 1.0f / jd.y, 1.0f / jd.z, 0.0f);
        }
        tab[2 * k] = t0;
        tab[2 * k + 1] = t1;
    }
}

// =========================================================================================
// Render kernel: main() of raytracingCs.glsl (:509-584), one lane per pixel.
// =========================================================================================
#ifndef RT_V0_WAVES
#define RT_V0_WAVES 0
#endif
#if RT_V0_WAVES > 0
#define RT_V0_BOUNDS __launch_bounds__(BLOCK_THREADS, RT_V0_WAVES)
#else
#define RT_V0_BOUNDS __launch_bounds__(BLOCK_THREADS)
#endif
template <int COUNT>
__global__ RT_V0_BOUNDS void rt_render_kernel(const RtFrame f, const RtDeviceScene dsc,
                                                                  float4 *__restrict__ gColor,
                                                                  float4 *__restrict__ gPosition,
                                                                  uint2 *__restrict__ gNormal,
                                                                  unsigned long long *rayCounter) {
    extern __shared__ float4 lds[];
    // ---- stage the compiled scene into LDS once per workgroup (coalesced 16-B copies)
    const int nF4 = f.nObj * (RT_HOT_F4 + RT_MAT_F4) + f.nLt * RT_LGT_F4 + 2 * (RT_HALTON_N / 4);
    for (int i = threadIdx.x; i < nF4; i += BLOCK_THREADS) lds[i] = dsc.compiled[i];
    __syncthreads();
    SceneLds sc;
    sc.hot = lds;
    sc.mat = sc.hot + f.nObj * RT_HOT_F4;
    sc.lgt = sc.mat + f.nObj * RT_MAT_F4;
    sc.halton2 = (const float *)(sc.lgt + f.nLt * RT_LGT_F4);
    sc.halton3 = sc.halton2 + RT_HALTON_N;

    // ---- lane -> pixel: wave w owns the 8x8 tile (w&1, w>>1) of the 16x16 workgroup tile
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * TILE + (wave & 1) * 8 + (lane & 7);
    const int j = blockIdx.y * TILE + (wave >> 1) * 8 + (lane >> 3);
    unsigned rays = 0;
    if (i < f.p.regionW && j < f.p.regionH) {
    const int gxI = f.p.x0 + i;
    const int ly = f.p.y0 + j;
    const int gyI = (ly / f.p.stripRows) * f.p.stripCycleRows + f.p.stripOffsetRows + ly % f.p.stripRows;
    const size_t outIdx = f.imageStores ? (size_t)gyI * f.p.width + gxI : (size_t)j * f.p.regionW + i;
    if (gxI >= f.p.width || gyI >= f.p.height) {   // outside the image: GL discards the imageStore
        if (!f.imageStores) {                      // (a whole-image surface has no such pixel)
            gColor[outIdx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            gPosition[outIdx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            gNormal[outIdx] = make_uint2(0u, 0u);
        }
    } else {
    const unsigned gx = (unsigned)gxI, gy = (unsigned)gyI;

    // jitter (:512-514): .x from the R8 sample, .y of an R8 texture is 0 -> -1 (A.1#3)
    const float nz = sample_noise(f, dsc.noise, gx, gy);
    const float jx = nz * 2.0f - 1.0f, jy = 0.0f * 2.0f - 1.0f;

    // generateCameraRay (:198-217)
    Ray ray;
    {
        float ux = (((float)gxI + 0.5f) + jx) / (float)f.p.width;
        float uy = (((float)gyI + 0.5f) + jy) / (float)f.p.height;
        ux = ux * 2.0f - 1.0f;
        uy = uy * 2.0f - 1.0f;
        ux *= f.sx;
        uy *= f.sy;
        v3 cd = V3(f.p.camDir[0], f.p.camDir[1], f.p.camDir[2]);
        v3 cr = V3(f.p.camRight[0], f.p.camRight[1], f.p.camRight[2]);
        v3 cu = V3(f.p.camUp[0], f.p.camUp[1], f.p.camUp[2]);
        ray.o = V3(f.p.camPos[0], f.p.camPos[1], f.p.camPos[2]);
        ray.d = normalize((cd + cr * ux) + cu * uy);
    }

    v3 finalColor = V3(0.0f, 0.0f, 0.0f), throughput = V3(1.0f, 1.0f, 1.0f);
    v3 P = V3(0.0f, 0.0f, 0.0f), N = V3(0.0f, 0.0f, 0.0f);   // undefined locals read as zero (A.3)

    for (int depth = 0; depth < f.p.maxRayDepth; ++depth) {
        float t;
        int idx = trace_closest<COUNT>(sc, f.nObj, ray, f.p.maxRayDistance, t, rays);
        if (idx < 0) {
            if (f.p.useSkybox && dsc.sky) finalColor = finalColor + throughput * sample_cube(dsc.sky, f.skySize, ray.d);
            // else `finalColor += throughput * vec3(0.0)` (:532): x*0.0 folds to 0.0 on the reference's
            // GL, so a NaN/inf throughput does not poison the colour on a miss (nan fixture).
            break;
        }
        // hit normal (:187-191): sphere = normalize(hit - centre); plane = raw normal
        {
            const float4 *h = sc.hot + idx * RT_HOT_F4;
            if (__float_as_int(h[0].w) == 0) N = normalize((ray.o + ray.d * t) - V3(h[2]));
            else N = V3(h[3]);
        }
        const Mat m = load_mat(sc, idx);
        P = ray.o + ray.d * t;
        v3 V = normalize(-ray.d);
        v3 Lo = compute_lighting<COUNT>(sc, f, P, N, m, V, nz, rays);
        finalColor = finalColor + throughput * Lo;

        if (depth > 2) {   // Russian roulette (:544-549)
            float dw = length(m.albedo) * m.diffuseStrength;
            float cp = fminf(fmaxf(throughput.x, fmaxf(throughput.y, throughput.z)) * 0.95f + dw, 0.99f);
            float rnd = random2((float)(gx + (unsigned)depth), (float)(gy + (unsigned)depth));
            if (rnd > cp) break;
            throughput = div3(throughput, cp);
        }
        float F = fresnel_schlick(fmaxf(dot(V, N), 0.0f), m.ior);
        if (m.diffuseStrength > 0.0f) {          // :555-567
            const int dd = depth < RT_MAX_DEPTH ? depth : RT_MAX_DEPTH - 1;
            v3 sd = reflect(ray.d, N);
            v3 hd = hemisphere_dir(V3(f.hemi[dd][0], f.hemi[dd][1], f.hemi[dd][2]), N);
            ray.d = normalize(mix_fast(sd, hd, m.roughness));
            ray.o = P + N * 0.001f;
            throughput = throughput * (m.albedo * m.diffuseStrength);
        } else if (m.transparency > 0.0f) {      // :568-571
            ray.d = calc_refraction(ray, N, m.ior);
            ray.o = P - N * 0.001f;
            throughput = throughput * ((m.albedo * (1.0f - F)) * m.transparency);
        } else {                                  // :572-576
            ray.d = reflect(ray.d, N);
            ray.o = P + N * 0.001f;
            throughput = throughput * (m.albedo * F);
        }
    }

    gColor[outIdx] = make_float4(finalColor.x, finalColor.y, finalColor.z, 1.0f);
    gPosition[outIdx] = make_float4(P.x, P.y, P.z, 1.0f);
    gNormal[outIdx] = make_uint2(f2h_rtz(N.x) | (f2h_rtz(N.y) << 16), f2h_rtz(N.z) | (0x3c00u << 16));

    }   // inside the image
    }   // inside the window

    if (COUNT) {   // instrumented build only: block-level sum in LDS, one global atomic per workgroup
        unsigned long long *blockRays = (unsigned long long *)(lds + nF4);
        if (threadIdx.x == 0) *blockRays = 0ull;
        __syncthreads();
        atomicAdd(blockRays, (unsigned long long)rays);
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(rayCounter, *blockRays);
    }
}

// =========================================================================================
// Render kernel, wavefront-packet variant (rt_packet.inc): same staging and lane->pixel map.
// =========================================================================================
// Instantiations.  One-wave workgroups (8x8 tile: the frame is a bag of independent wave-sized jobs, no intra-group
// imbalance) for every scene the ABI accepts (<= RT_MAX_OBJECTS).  Wave-uniform records (the candidate being tested, the
// current light, Halton entries) come by scalar loads from the global copy of the compiled scene, the hit object's shape /
// material fields by per-lane loads from it; LDS holds the per-lane parking area and, in the LIGHT profile, the AABBs.
// Profiles: rt_packet.inc.  Measured on one MI355X, ms/frame at full size (gpurun_out/try8..19.log):
//                                                     C2       C3      C4      C5
//   round 1 shapes (whole scene in LDS, 4 waves/SIMD)  0.459    5.27    8.25    48.4   (after the fast 1/x, sqrt paths)
//   LIGHT / HEAVY profiles                             0.404    4.84    8.15    43.2
#ifndef RT_PK_WAVES_SMALL
#define RT_PK_WAVES_SMALL 5
#endif
#ifndef RT_PK_LIGHT_SCENE
#define RT_PK_LIGHT_SCENE 32    // objects: at or below, the LIGHT profile
#endif
template <int COUNT, int BT, bool COMPACT, typename PROFILE>
__global__ __launch_bounds__(BT, PROFILE::waves)
void rt_render_packet_kernel(const RtFrame f, const RtDeviceScene dsc, float4 *__restrict__ gColor,
                             float4 *__restrict__ gPosition, uint2 *__restrict__ gNormal,
                             unsigned long long *rayCounter) {
    extern __shared__ float4 lds[];
    const int nAll = f.nObj * (RT_HOT_F4 + RT_MAT_F4) + f.nLt * RT_LGT_F4 + 2 * (RT_HALTON_N / 4);
    const int nF4 = !PROFILE::boundsLds ? 0 : (COMPACT ? f.nObj * 2 : nAll);          // float4 staged in LDS
    if (COMPACT) {
        for (int i = threadIdx.x; i < nF4; i += BT) lds[i] = dsc.compiled[(i >> 1) * RT_HOT_F4 + (i & 1)];
    } else {
        for (int i = threadIdx.x; i < nF4; i += BT) lds[i] = dsc.compiled[i];
    }
    __syncthreads();
    SceneLds sc;
    sc.compact = COMPACT;
    sc.keepAabb = PROFILE::keepAabb;
    sc.boundsLds = PROFILE::boundsLds;
    sc.straight = PROFILE::straight;
    sc.split = PROFILE::split;
    sc.wedge = PROFILE::wedge;
    sc.hotStride = COMPACT ? 2 : RT_HOT_F4;
    sc.matF4Base = f.nObj * RT_HOT_F4;
    sc.hot = lds;
    sc.mat = sc.hot + f.nObj * RT_HOT_F4;                  // (not staged, never read through LDS, when COMPACT)
    sc.lgt = sc.mat + f.nObj * RT_MAT_F4;
    sc.halton2 = (const float *)(sc.lgt + f.nLt * RT_LGT_F4);
    sc.halton3 = sc.halton2 + RT_HALTON_N;
    sc.stats = COUNT ? rayCounter : nullptr;
    sc.global = dsc.compiled;
    sc.lgtF4Base = f.nObj * (RT_HOT_F4 + RT_MAT_F4);
    sc.haltonFloatBase = (sc.lgtF4Base + f.nLt * RT_LGT_F4) * 4;
    sc.park = (float *)(lds + nF4 + 1);      // after the staged scene and the 16-byte counter slot
    sc.pcfTabF4 = dsc.noise ? -1 : nAll;     // the tables follow the staged sections in the global copy
    sc.stab = dsc.shadowTab;

    // tile of this workgroup: longest-first order from the previous frame's measured costs, if any
    constexpr int TILE_ = (BT == 256) ? 16 : 8;
    const int tilesX = (f.p.regionW + TILE_ - 1) / TILE_;
    const unsigned tile = dsc.tileOrder ? dsc.tileOrder[blockIdx.x] : blockIdx.x;
    // an order buffer is a permutation of [0, gridDim.x) by construction (identity-initialised, rewritten only by
    // rt_lpt_sort_kernel); the guard keeps a corrupted entry from turning into out-of-bounds tileCost / surface writes
    if (tile >= gridDim.x) return;
    sc.tileX = (int)(tile % (unsigned)tilesX);
    sc.tileY = (int)(tile / (unsigned)tilesX);
    const long long t0 = clock64();

    unsigned rays = 0;
    render_packet<COUNT, BT, PROFILE>(f, dsc, sc, gColor, gPosition, gNormal, rays);

    if (dsc.tileCost && (threadIdx.x & 63) == 0) {     // one add per wave: tile cost = sum of its waves' cycles / 64
        const unsigned c = (unsigned)(((unsigned long long)(clock64() - t0)) >> 6);
#ifndef RT_FB_ACCUM
#define RT_FB_ACCUM 1
#endif
        // costs accumulate between two sorts: the order then follows the tiles' average cost over the last period, which is what predicts the next
        // frames when frameCount rotates the shared bounce sample from frame to frame
        // (atomic: frames in flight on several streams may add to the same tile at once -- ADVICE r2)
        if (RT_FB_ACCUM) atomicAdd(&dsc.tileCost[tile], c);
        else dsc.tileCost[tile] = c;
    }

    if (COUNT) {
        unsigned long long *blockRays = (unsigned long long *)(lds + nF4);
        if (threadIdx.x == 0) *blockRays = 0ull;
        __syncthreads();
        atomicAdd(blockRays, (unsigned long long)rays);
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(rayCounter, *blockRays);
    }
}

// =========================================================================================
// Tile-cost PREDICTOR: the heavy-first workgroup order of a frame from that frame's own inputs.
// =========================================================================================
// A frame is a bag of one-wave tiles whose costs differ by an order of magnitude (a wall tile ends after one bounce, a tile on
// a glass sphere above the floor runs every bounce, light and sample), and a tile that starts late ends late: in raster order
// the frame takes 30 % longer than with the heavy tiles first.  Rounds 1-2 ordered the tiles by their MEASURED cost of the
// previous frames, which only helps while consecutive frames are the same.  This kernel needs no history: one lane per tile
// follows the path of the tile's centre pixel -- closest hits by the exact tests, the reference's bounce rule (:552-576), no
// shadow rays, no roulette -- and prices it: per bounce the traversal, per light the set-up and BRDF, per shadow ray its set-up
// plus one candidate test for every object the light's shadow table lists for that shading point (rt_shadowtab.inc: the very
// objects the render kernel will visit), 16 blocker rays for a PCSS light, 4 probes on a subsurface material.  The costs go
// into 32 geometric classes (ratio 2^(1/4)); rt_scatter_tiles_kernel writes the tiles class by class, heaviest first.
// Scheduling only: no pixel depends on the order (the tests run raster, predicted and measured orders against the oracle).
// One lane per tile (its centre pixel), one kernel: a lane appends its tile to its class's segment (seg[class][...], one
// wave-aggregated atomic per class present in the wave), and the render kernel maps its workgroup id to (class, rank) from
// the 32 class sizes -- no counting pass, no sort.  cursors = this launch's 32 class sizes (zero on entry), nextCursors =
// the other set, cleared here for the next prediction on this stream.  Candidate tests per shadow ray are priced with a
// scene-wide constant (a table gather per light and bounce would double the pass's latency for little extra order).
__global__ __launch_bounds__(64) void rt_predict_tiles_kernel(const RtFrame f, const RtDeviceScene dsc, int tilesX, int nTiles,
                                                               unsigned *__restrict__ seg, int segStride, unsigned *__restrict__ cursors,
                                                               unsigned *__restrict__ nextCursors, unsigned char *__restrict__ cls) {
    const int lane = threadIdx.x;
    const int tile = blockIdx.x * 64 + lane;        // (consecutive lanes = consecutive tiles of a row: inside a class the tiles stay in
    const bool valid = tile < nTiles;               //  raster order -- a strided walk over the image measured 7-19 % SLOWER, neighbours share cache lines)
    if (blockIdx.x == 0 && lane < 32) nextCursors[lane] = 0u;
    // the objects' hot records in LDS (broadcast reads pipeline; dependent scalar loads at two waves per SIMD do not: 36 -> 10 us)
    extern __shared__ float4 lds[];
    for (int k = lane; k < f.nObj * RT_HOT_F4; k += 64) lds[k] = dsc.compiled[k];
    __syncthreads();
    SceneLds sc;
    sc.hot = lds;
    sc.mat = dsc.compiled + f.nObj * RT_HOT_F4;
    sc.lgt = sc.mat + f.nObj * RT_MAT_F4;
    sc.global = dsc.compiled;
    sc.compact = true;
    sc.matF4Base = f.nObj * RT_HOT_F4;
    sc.lgtF4Base = f.nObj * (RT_HOT_F4 + RT_MAT_F4);
    const int tx = valid ? tile % tilesX : 0, ty = valid ? tile / tilesX : 0;
    const int i = min(tx * 8 + 4, f.p.regionW - 1), j = min(ty * 8 + 4, f.p.regionH - 1);
    const int gxI = f.p.x0 + i, ly = f.p.y0 + j;
    const int gyI = (ly / f.p.stripRows) * f.p.stripCycleRows + f.p.stripOffsetRows + ly % f.p.stripRows;
    float cost = 300.0f;                 // ray generation, stores
    bool alive = valid && gxI < f.p.width && gyI < f.p.height;
    const float nz = alive ? sample_noise(f, dsc.noise, (unsigned)gxI, (unsigned)gyI) : 0.0f;
    Ray ray;
    {
        float ux = (((float)gxI + 0.5f) + (nz * 2.0f - 1.0f)) / (float)f.p.width, uy = (((float)gyI + 0.5f) - 1.0f) / (float)f.p.height;
        ux = (ux * 2.0f - 1.0f) * f.sx;
        uy = (uy * 2.0f - 1.0f) * f.sy;
        const v3 cd = V3(f.p.camDir[0], f.p.camDir[1], f.p.camDir[2]), cr = V3(f.p.camRight[0], f.p.camRight[1], f.p.camRight[2]),
                 cu = V3(f.p.camUp[0], f.p.camUp[1], f.p.camUp[2]);
        ray.o = V3(f.p.camPos[0], f.p.camPos[1], f.p.camPos[2]);
        ray.d = normalize((cd + cr * ux) + cu * uy);
    }
    const float travCost = 250.0f + 12.0f * (float)min(f.nObj, 64) + 2.0f * (float)f.nObj;      // one packet traversal (masks + a few candidates)
    const float cand = 1.5f + (float)f.nObj * (1.0f / 28.0f);                                     // objects a shading point's table cell lists, typically
    for (int depth = 0; depth < f.p.maxRayDepth; ++depth) {
        if (__builtin_amdgcn_ballot_w64(alive) == 0ull) break;
        // closest hit: every object, the exact tests
        float t = f.p.maxRayDistance;
        int idx = -1;
        {
            v3 inv;
            inv.x = __builtin_amdgcn_rcpf(ray.d.x); inv.y = __builtin_amdgcn_rcpf(ray.d.y); inv.z = __builtin_amdgcn_rcpf(ray.d.z);
            const float a = dot(ray.d, ray.d);
            for (int k = 0; k < f.nObj; k++) {
                const float4 *h = lds + k * RT_HOT_F4;
                const float4 h0 = h[0], h1 = h[1];
                if (alive && aabb_test(ray, inv, h0, h1, f.p.maxRayDistance)) {
                    float tt;
                    if (shape_test(ray, a, h, __float_as_int(h0.w), tt) && tt > 0.0f && tt < t) { t = tt; idx = k; }
                }
            }
        }
        if (alive) cost += travCost;
        alive = alive && idx >= 0;
        const int ii = alive ? idx : 0;
        const float4 hb = lds[ii * RT_HOT_F4], h2 = lds[ii * RT_HOT_F4 + 2], h3 = lds[ii * RT_HOT_F4 + 3];
        const float4 m1 = pk_lane_mat(sc, ii, 1), m2 = pk_lane_mat(sc, ii, 2);
        const v3 P = ray.o + ray.d * t;
        const v3 N = (__float_as_int(hb.w) == 0) ? normalize(P - V3(h2)) : V3(h3);
        if (alive) cost += 350.0f;       // hit set-up, parking, next direction
        for (int li = 0; li < f.nLt; li++) {
            const int lb4 = sc.lgtF4Base + li * RT_LGT_F4;
            const float4 l0 = uni_load4(sc, lb4), l1 = uni_load4(sc, lb4 + 1), l3 = uni_load4(sc, lb4 + 3);
            const int ltype = __float_as_int(l0.w), shadowType = __float_as_int(l3.x), pcf = max(__float_as_int(l3.y), 0);
            v3 lightDir = V3(l1);
            bool lit = alive;
            if (ltype != 1) {
                lightDir = V3(l0) - P;
                lightDir = lightDir * __builtin_amdgcn_rsqf(dot(lightDir, lightDir));
                if (ltype == 2) lit = lit && dot(lightDir, V3(l1)) > 0.0f;
            }
            lit = lit && dot(N, lightDir) > 0.0f;
            if (alive) cost += 330.0f;   // light set-up + computePBR
            if (lit && (shadowType == 1 || shadowType == 2)) {
                cost += 120.0f + (float)min(pcf, 64) * (50.0f + 70.0f * cand);
                if (shadowType == 2) cost += 16.0f * (60.0f + 25.0f * (float)min(f.nObj, 32));
            }
        }
        if (alive && m2.w > 0.0f) cost += 4.0f * (travCost + 60.0f);
        if (depth + 1 >= f.p.maxRayDepth) break;
        if (alive) {
            const v3 O = P + N * 0.001f;
            if (m1.y > 0.0f) {
                const int dd = depth < RT_MAX_DEPTH ? depth : RT_MAX_DEPTH - 1;
                ray.d = normalize(mix_fast(reflect(ray.d, N), hemisphere_dir(V3(f.hemi[dd][0], f.hemi[dd][1], f.hemi[dd][2]), N), m1.x));
                ray.o = O;
            } else if (m1.w > 0.0f) {
                ray.d = calc_refraction(ray, N, m1.z);
                ray.o = P - N * 0.001f;
            } else {
                ray.d = reflect(ray.d, N);
                ray.o = O;
            }
        }
    }
    // 32 geometric classes, ratio 2^(1/4), from 2^8 = 256 units; NaN / inf -> top class
    cost = (cost == cost) ? fminf(cost, 3.0e38f) : 3.0e38f;
    int c = (int)(4.0f * __log2f(fmaxf(cost, 1.0f))) - 32;
    c = min(max(c, 0), 31);
    // a tile next to a heavier one is probably cut by the same silhouette: pull it up to one class below its row neighbours
    {
        const int cl = __shfl_up(c, 1), cr2 = __shfl_down(c, 1);
        const int nb = max((lane & 63) > 0 ? cl : 0, (lane & 63) < 63 ? cr2 : 0);
        c = max(c, nb - 1);
    }
    if (valid && cls) cls[tile] = (unsigned char)c;
    unsigned long long rem = __builtin_amdgcn_ballot_w64(valid);
    while (rem) {                        // one atomic per class present in the wave
        const int c0 = __builtin_amdgcn_readlane(c, __builtin_ctzll(rem));
        const bool sel = valid && c == c0;
        const unsigned long long mm = __builtin_amdgcn_ballot_w64(sel);
        const int first = __builtin_ctzll(mm);
        unsigned base = 0;
        if (lane == first) base = atomicAdd(&cursors[c0], (unsigned)__builtin_popcountll(mm));
        base = (unsigned)__builtin_amdgcn_readlane((int)base, first);
        if (sel) {
            const unsigned pos = base + (unsigned)__builtin_popcountll(mm & ((1ull << lane) - 1ull));
            if (pos < (unsigned)segStride) seg[(size_t)c0 * segStride + pos] = (unsigned)tile;
        }
        rem &= ~mm;
    }
}

// =========================================================================================
// Rank-0 reassembly of gathered interleaved strips (multi-GPU): pure copy kernel, 16 B/lane.
// =========================================================================================
template <typename U>
__global__ void rt_deinterleave_kernel(const U *__restrict__ src, U *__restrict__ dst, int rowUnits,
                                       int height, int stripRows, int stripCount, size_t rankStrideUnits) {
    const size_t total = (size_t)rowUnits * height;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (size_t)gridDim.x * blockDim.x) {
        int y = (int)(k / rowUnits), c = (int)(k % rowUnits);
        int strip = y / stripRows;
        int rank = strip % stripCount, localStrip = strip / stripCount;
        int ly = localStrip * stripRows + y % stripRows;
        dst[k] = src[(size_t)rank * rankStrideUnits + (size_t)ly * rowUnits + c];
    }
}

// =========================================================================================
// Wire format of the gather (30 B/pixel: rgb f32 | rgb f32 | rgb f16; alpha is the constant 1.0 on all
// three surfaces): pack on every rank, unpack + de-interleave on rank 0.  Streaming copies: the
// 12-byte-per-lane accesses of a wave are contiguous (768 B), the 6-byte ones 384 B.
// =========================================================================================
__global__ __launch_bounds__(256) void rt_wire_pack_kernel(const float4 *__restrict__ col, const float4 *__restrict__ pos,
                                                           const uint2 *__restrict__ nrm, float *__restrict__ wcol,
                                                           float *__restrict__ wpos, unsigned short *__restrict__ wnrm, size_t n) {
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
        const float4 c = col[k], q = pos[k];
        const uint2 h = nrm[k];
        wcol[3 * k] = c.x; wcol[3 * k + 1] = c.y; wcol[3 * k + 2] = c.z;
        wpos[3 * k] = q.x; wpos[3 * k + 1] = q.y; wpos[3 * k + 2] = q.z;
        wnrm[3 * k] = (unsigned short)(h.x & 0xffffu); wnrm[3 * k + 1] = (unsigned short)(h.x >> 16);
        wnrm[3 * k + 2] = (unsigned short)(h.y & 0xffffu);
    }
}

// Rank-0 side.  Every cycle of rootRows + (stripCount-1)*stripRows image rows starts with the root's own rows
// -- copied straight from its local rgba surfaces when rootCol != NULL (they never travel), otherwise taken
// from wire slot 0 like everybody else's -- followed by one strip of each peer, from that peer's wire buffer.
__global__ __launch_bounds__(256) void rt_wire_unpack_kernel(const unsigned char *__restrict__ wire, size_t rankStrideBytes,
                                                             size_t rankPixels, const float4 *__restrict__ rootCol,
                                                             const float4 *__restrict__ rootPos, const uint2 *__restrict__ rootNrm,
                                                             int rootRows, float4 *__restrict__ col, float4 *__restrict__ pos,
                                                             uint2 *__restrict__ nrm, int width, int height, int stripRows,
                                                             int stripCount) {
    const size_t total = (size_t)width * height;
    const int cycleRows = rootRows + (stripCount - 1) * stripRows;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (size_t)gridDim.x * 256) {
        const int y = (int)(k / width), x = (int)(k % width);
        const int cyc = y / cycleRows, w = y % cycleRows;
        int rank, ly;
        if (w < rootRows) {
            rank = 0;
            ly = cyc * rootRows + w;
            if (rootCol) {
                const size_t pi = (size_t)ly * width + x;
                col[k] = rootCol[pi];
                pos[k] = rootPos[pi];
                nrm[k] = rootNrm[pi];
                continue;
            }
        } else {
            const int j = w - rootRows;
            rank = 1 + j / stripRows;
            ly = cyc * stripRows + j % stripRows;
        }
        const size_t pi = (size_t)ly * width + x;
        const unsigned char *base = wire + (size_t)rank * rankStrideBytes;
        const float *wc = (const float *)base + 3 * pi;
        const float *wp = (const float *)(base + rankPixels * 12) + 3 * pi;
        const unsigned short *wn = (const unsigned short *)(base + rankPixels * 24) + 3 * pi;
        col[k] = make_float4(wc[0], wc[1], wc[2], 1.0f);
        pos[k] = make_float4(wp[0], wp[1], wp[2], 1.0f);
        nrm[k] = make_uint2((unsigned)wn[0] | ((unsigned)wn[1] << 16), (unsigned)wn[2] | (0x3c00u << 16));
    }
}

// =========================================================================================
// Launch wrappers
// =========================================================================================
hipError_t rt_launch_compile_scene(const uint8_t *dObjects, int nObj, const uint8_t *dLights, int nLt,
                                   float4 *dCompiled, hipStream_t s) {
    hipLaunchKernelGGL(rt_compile_scene_kernel, dim3(1), dim3(256), 0, s, dObjects, nObj, dLights, nLt, dCompiled);
    return hipGetLastError();
}

// Shadow tables of the current compiled scene (rt_shadowtab.inc): headers by one workgroup, then one thread per cell.
#ifndef RT_ST_NMAX
#define RT_ST_NMAX 8.0f         // lanes whose shading normal is longer take every object (the tables' aim-error bound assumes |N| <= this)
#endif
hipError_t rt_launch_shadow_tables(const float4 *dCompiled, int nObj, int nLt, unsigned *dTab, const RtShadowTabGeom &g, hipStream_t s, bool onePhase,
                                   bool blocker) {
    if (nLt <= 0 || nObj <= 0 || nObj > RT_ST_MAX_OBJECTS || nLt > RT_ST_MAX_LIGHTS) return hipSuccess;
    const int NW = rt_shadowtab_words(nObj);
    const size_t cube = (size_t)g.NB * 6 * g.Kcube * g.Kcube, plan = (size_t)g.NB * g.Kplan * g.Kplan + 1;
    hipLaunchKernelGGL(rt_shadowtab_headers_kernel, dim3((unsigned)nLt), dim3(64), 0, s, dCompiled, nObj, nLt, (float4 *)dTab, g.Kcube, g.Kplan,
                       g.NB, NW, RT_ST_NMAX, (unsigned)(cube > plan ? cube : plan), blocker ? 1 : 0);
    const size_t maxCells = cube > plan ? cube : plan, dirCells = rt_shadowtab_dir_cells(g), dirBase = rt_shadowtab_table_dwords(g, nObj, nLt, blocker);
    const dim3 gridCells((unsigned)((maxCells + 255) / 256), (unsigned)nLt), gridDir((unsigned)((dirCells + 255) / 256), (unsigned)nLt, (unsigned)NW);
    for (int b = 0; b < (blocker ? 2 : 1); b++) {        // the PCF tables, then (scenes with a PCSS light) the blocker tables
        const size_t scratch = dirBase + (size_t)b * nLt * dirCells * NW;
        if (onePhase) {
            hipLaunchKernelGGL(rt_shadowtab_build_kernel<0>, gridCells, dim3(256), 0, s, dCompiled, nObj, nLt, (const float4 *)dTab, dTab, NW, scratch, dirCells, b);
        } else {      // every object per direction cell once, then the per-bin tests for the survivors (rt_shadowtab.inc)
            hipLaunchKernelGGL(rt_shadowtab_build_kernel<1>, gridDir, dim3(256), 0, s, dCompiled, nObj, nLt, (const float4 *)dTab, dTab, NW, scratch, dirCells, b);
            hipLaunchKernelGGL(rt_shadowtab_build_kernel<2>, gridCells, dim3(256), 0, s, dCompiled, nObj, nLt, (const float4 *)dTab, dTab, NW, scratch, dirCells, b);
        }
    }
    return hipGetLastError();
}

// countMode (only with dRayCounter): 1 = instrumented build that traces every ray the REFERENCE traces (its count is the
// unit count R of the metric), 2 = instrumented build that keeps the production kernel's provably-dead-ray skips
// (its count is what the timed kernel actually traverses).
hipError_t rt_launch_render(const RtFrame &f, const RtDeviceScene &sc, float4 *dColor, float4 *dPos,
                            uint2 *dNormal, unsigned long long *dRayCounter, int variant, hipStream_t s, int countMode) {
    if (f.p.regionW <= 0 || f.p.regionH <= 0) return hipSuccess;
    if (variant != 1 && (f.nObj > RT_EXHAUSTIVE_MAX_OBJECTS || f.nLt > RT_EXHAUSTIVE_MAX_LIGHTS)) return hipErrorInvalidValue;      // (the ABI refuses first)
    const size_t sceneBytes = (rt_compiled_f4(f.nObj, f.nLt) + 1) * sizeof(float4);   // exhaustive kernel: whole scene + 16 B block counter
    if (variant == 1) {
        int bt, tile, tilesX, nTiles;
        rt_packet_geometry(f.nObj, f.p.regionW, f.p.regionH, &bt, &tile, &tilesX, &nTiles);
        dim3 grid(nTiles);
        const bool light = bt == 64 && f.nObj <= RT_PK_LIGHT_SCENE && !(f.nObj > 0 && f.nLt > 0 && !sc.shadowTab);
        // LDS: the AABBs (2 float4 per object) + the 16-byte counter slot + the profile's parking area
        const size_t ldsBytes = ((size_t)((light || PkHeavy::boundsLds) ? f.nObj * 2 : 0) + 1) * sizeof(float4) +
                                (size_t)(light ? PkLight::parkFloats : PkHeavy::parkFloats) * bt * sizeof(float);
#define RT_LAUNCH_PK(BT_, PROFILE_)                                                                                                   \
        do {                                                                                                                          \
            if (dRayCounter && countMode == 2) hipLaunchKernelGGL((rt_render_packet_kernel<2, BT_, true, PROFILE_>), grid, dim3(BT_), ldsBytes, s, f, sc, dColor, dPos, dNormal, dRayCounter); \
            else if (dRayCounter) hipLaunchKernelGGL((rt_render_packet_kernel<1, BT_, true, PROFILE_>), grid, dim3(BT_), ldsBytes, s, f, sc, dColor, dPos, dNormal, dRayCounter);            \
            else hipLaunchKernelGGL((rt_render_packet_kernel<0, BT_, true, PROFILE_>), grid, dim3(BT_), ldsBytes, s, f, sc, dColor, dPos, dNormal, dRayCounter);                             \
        } while (0)
        // scenes without shadow tables (more than RT_ST_MAX_OBJECTS objects or RT_ST_MAX_LIGHTS lights: rt_set_scene builds none)
        // run the profile that finds the lights' candidates per packet
        const bool noTab = PkLight::tabWords > 0 && f.nObj > 0 && f.nLt > 0 && (!sc.shadowTab || f.nObj > RT_ST_MAX_OBJECTS);
        if (f.anyPcss) {
            if (noTab) RT_LAUNCH_PK(64, PkHugeS);
            else if (light) RT_LAUNCH_PK(64, PkLightS);
            else if (f.nObj <= 64) RT_LAUNCH_PK(64, PkHeavy1S);
            else RT_LAUNCH_PK(64, PkHeavyS);
        } else {
            if (noTab) RT_LAUNCH_PK(64, PkHuge);
            else if (light) RT_LAUNCH_PK(64, PkLight);
            else if (f.nObj <= 64) RT_LAUNCH_PK(64, PkHeavy1);
            else RT_LAUNCH_PK(64, PkHeavy);
        }
#undef RT_LAUNCH_PK
    } else {
        dim3 grid((f.p.regionW + TILE - 1) / TILE, (f.p.regionH + TILE - 1) / TILE);
        if (dRayCounter)
            hipLaunchKernelGGL(rt_render_kernel<1>, grid, dim3(BLOCK_THREADS), sceneBytes, s, f, sc, dColor, dPos, dNormal, dRayCounter);
        else
            hipLaunchKernelGGL(rt_render_kernel<0>, grid, dim3(BLOCK_THREADS), sceneBytes, s, f, sc, dColor, dPos, dNormal, dRayCounter);
    }
    return hipGetLastError();
}

// Predicted heavy-first tile order of frame f into dOrder (nTiles entries).  Scratch: dSeg = 32 x segStride tile ids (the tiles of
// class c), dCursors = 32 class sizes (zero on entry; dNextCursors is cleared for the next prediction); dCls (may be NULL) = class
// per tile for diagnostics.
__global__ __launch_bounds__(256) void rt_compact_order_kernel(const unsigned *__restrict__ seg, int segStride, const unsigned *__restrict__ counts,
                                                                unsigned *__restrict__ order, int nTiles) {
    __shared__ unsigned off[33];
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int c = 31; c >= 0; c--) { off[c] = run; run += counts[c]; }      // heaviest class first
        off[32] = run;
    }
    __syncthreads();
    for (int c = 31; c >= 0; c--) {
        const unsigned n = counts[c];
        for (unsigned k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
            const unsigned pos = off[c] + k;
            if (pos < (unsigned)nTiles) order[pos] = seg[(size_t)c * segStride + k];
        }
    }
}

hipError_t rt_launch_predict_order(const RtFrame &f, const RtDeviceScene &sc, int tilesX, int nTiles, unsigned *dSeg, int segStride,
                                   unsigned *dCursors, unsigned *dNextCursors, unsigned char *dCls, unsigned *dOrder, hipStream_t s) {
    if (nTiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(rt_predict_tiles_kernel, dim3((unsigned)((nTiles + 63) / 64)), dim3(64), (size_t)f.nObj * RT_HOT_F4 * sizeof(float4), s, f,
                       sc, tilesX, nTiles, dSeg, segStride, dCursors, dNextCursors, dCls);
    // the class segments as one order array (the render kernel reads tileOrder[blockIdx.x] and nothing else)
    hipLaunchKernelGGL(rt_compact_order_kernel, dim3((unsigned)((nTiles + 255) / 256)), dim3(256), 0, s, dSeg, segStride, dCursors, dOrder, nTiles);
    return hipGetLastError();
}

void rt_packet_geometry(int nObj, int regionW, int regionH, int *bt, int *tile, int *tilesX, int *nTiles) {
    (void)nObj;         // one-wave workgroups / 8x8-pixel tiles for every scene size (the 256-thread shape is retired)
    *bt = 64;
    *tile = 8;
    *tilesX = (regionW + *tile - 1) / *tile;
    *nTiles = *tilesX * ((regionH + *tile - 1) / *tile);
}

// Longest-processing-time-first order from measured tile costs: 256 linear cost bins (descending),
// counting sort by one 1024-thread workgroup (a frame has at most a few 1e5 tiles).  Order inside a
// bin is arbitrary -- this is a scheduling hint, never a data dependency.  Frames on other streams may
// be updating `cost` while this runs, so every cost is read ONCE into `snap` and both passes bin the
// snapshot: `order` is always a permutation of the tiles.  Costs are cleared for the next accumulation.
// `accum` (optional): the costs read here are also added to a second cost buffer (the per-phase sorts of a free-running
// frameCount feed the all-phase average this way, rt_abi.cpp).
__global__ __launch_bounds__(1024) void rt_lpt_sort_kernel(unsigned *__restrict__ cost, unsigned *__restrict__ snap,
                                                           unsigned *__restrict__ order, int nTiles, unsigned *__restrict__ accum) {
    __shared__ unsigned bins[256];
    __shared__ unsigned maxCost;
    if (threadIdx.x < 256) bins[threadIdx.x] = 0;
    if (threadIdx.x == 0) maxCost = 1;
    __syncthreads();
    unsigned mx = 1;
    for (int i = threadIdx.x; i < nTiles; i += 1024) {
        const unsigned c = __hip_atomic_load(&cost[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        snap[i] = c;
        if (accum) atomicAdd(&accum[i], c);
        mx = max(mx, c);
    }
    atomicMax(&maxCost, mx);
    __syncthreads();
    const float scale = 255.0f / (float)maxCost;
    for (int i = threadIdx.x; i < nTiles; i += 1024) {      // each thread re-reads only what it wrote itself
        const unsigned b = 255u - (unsigned)min(255, (int)((float)snap[i] * scale));
        snap[i] = b;
        atomicAdd(&bins[b], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int b = 0; b < 256; b++) { unsigned n = bins[b]; bins[b] = run; run += n; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nTiles; i += 1024) {
        const unsigned pos = atomicAdd(&bins[snap[i]], 1u);
        order[pos] = (unsigned)i;
    }
    for (int i = threadIdx.x; i < nTiles; i += 1024) cost[i] = 0;
}

__global__ void rt_iota_kernel(unsigned *__restrict__ order, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[i] = (unsigned)i;
}

#if RT_FASTMATH_STATS
extern "C" int rt_debug_fastmath_fallbacks(unsigned long long out[4], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rtf::g_fallbacks), 32) != hipSuccess) return -3;
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rtf::g_fallbacks), z, 32) != hipSuccess) return -3; }
    return 0;
}
#endif

hipError_t rt_launch_iota(unsigned *dOrder, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(rt_iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dOrder, n);
    return hipGetLastError();
}

hipError_t rt_launch_lpt_sort(unsigned *dCost, unsigned *dSnap, unsigned *dOrder, int nTiles, hipStream_t s, unsigned *dAccum) {
    if (nTiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(rt_lpt_sort_kernel, dim3(1), dim3(1024), 0, s, dCost, dSnap, dOrder, nTiles, dAccum);
    return hipGetLastError();
}

hipError_t rt_launch_deinterleave(const void *src, void *dst, int width, int height, int bytesPerPixel,
                                  int stripRows, int stripCount, size_t rankStrideBytes, hipStream_t s) {
    size_t rowBytes = (size_t)width * bytesPerPixel;
    // widest unit that divides a row, the rank stride and both base addresses: 16 B/lane for
    // the rgba32f surfaces (and rgba16f at even widths)
    size_t all = rowBytes | rankStrideBytes | (size_t)(uintptr_t)src | (size_t)(uintptr_t)dst;
    int unit = (all % 16 == 0) ? 16 : (all % 8 == 0) ? 8 : (all % 4 == 0) ? 4 : 0;
    if (!unit) return hipErrorInvalidValue;
    int rowUnits = (int)(rowBytes / unit);
    size_t strideUnits = rankStrideBytes / unit;
    size_t total = (size_t)rowUnits * height;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    if (unit == 16)
        hipLaunchKernelGGL(rt_deinterleave_kernel<uint4>, dim3(blocks), dim3(256), 0, s, (const uint4 *)src, (uint4 *)dst,
                           rowUnits, height, stripRows, stripCount, strideUnits);
    else if (unit == 8)
        hipLaunchKernelGGL(rt_deinterleave_kernel<uint2>, dim3(blocks), dim3(256), 0, s, (const uint2 *)src, (uint2 *)dst,
                           rowUnits, height, stripRows, stripCount, strideUnits);
    else
        hipLaunchKernelGGL(rt_deinterleave_kernel<unsigned>, dim3(blocks), dim3(256), 0, s, (const unsigned *)src,
                           (unsigned *)dst, rowUnits, height, stripRows, stripCount, strideUnits);
    return hipGetLastError();
}

hipError_t rt_launch_wire_pack(const void *dColor, const void *dPos, const void *dNormal, void *dWire, size_t nPixels,
                               hipStream_t s) {
    if (nPixels == 0) return hipSuccess;
    size_t blocks = (nPixels + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    unsigned char *w = (unsigned char *)dWire;
    hipLaunchKernelGGL(rt_wire_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float4 *)dColor, (const float4 *)dPos,
                       (const uint2 *)dNormal, (float *)w, (float *)(w + nPixels * 12), (unsigned short *)(w + nPixels * 24), nPixels);
    return hipGetLastError();
}

hipError_t rt_launch_wire_unpack(const void *dWire, size_t rankStrideBytes, size_t rankPixels, const void *dRootColor,
                                 const void *dRootPos, const void *dRootNormal, int rootRows, void *dColor, void *dPos,
                                 void *dNormal, int width, int height, int stripRows, int stripCount, hipStream_t s) {
    size_t blocks = ((size_t)width * height + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(rt_wire_unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned char *)dWire, rankStrideBytes,
                       rankPixels, (const float4 *)dRootColor, (const float4 *)dRootPos, (const uint2 *)dRootNormal, rootRows,
                       (float4 *)dColor, (float4 *)dPos, (uint2 *)dNormal, width, height, stripRows, stripCount);
    return hipGetLastError();
}
