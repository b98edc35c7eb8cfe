/*
 * rt_oracle.c -- TEST INFRASTRUCTURE (see rt_oracle.h).  Scalar fp32 restatement of
 * /root/reference/shader/raytracingCs.glsl, one function per GLSL function, each
 * citing the lines it follows.  Evaluation orders follow the llvmpipe lowering rules
 * probed in SURVEY.md Appendix A.3 (no FMA contraction, dot = (z*z + y*y) + x*x,
 * normalize = v * (1/sqrt(dot)), mix = a + t*(b-a), true IEEE divides, ...).
 *
 * Must be compiled with -ffp-contract=off and without -ffast-math (oracle/Makefile).
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const float PI_F = 3.14159265359f; /* :6 */

typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 div3(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 divs3(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline v3 splat3(float s) { return V3(s, s, s); }
/* A.3: dot(a,b) = ((a.z*b.z + a.y*b.y) + a.x*b.x) */
static inline float dot3(v3 a, v3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
/* A.3: normalize(a) = a * (1.0/sqrt(dot(a,a))) */
static inline v3 normalize3(v3 a) { return scale3(a, 1.0f / sqrtf(dot3(a, a))); }
static inline v3 cross3(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* mix(a,b,t).  Mesa lowers the built-in context-dependently (nir_lower_flrp); probed bitwise
 * on llvmpipe in the two shapes the shader uses (tests/test_oracle_units.py, fixture tests/golden/probes.npz):
 *   - all-variable operands, t used by no other mix  ->  a + t*(b-a)        (:562)
 *   - constant first operand (vec3(0.04))             ->  a*(1-t) + b*t      (:240) */
static inline v3 mix3_fast(v3 a, v3 b, float t) { return add3(a, scale3(sub3(b, a), t)); }
static inline v3 mix3_strict(v3 a, v3 b, float t) { return add3(scale3(a, 1.0f - t), scale3(b, t)); }
/* A.3: reflect(I,N) = I - (2*dot(N,I))*N */
static inline v3 reflect3(v3 I, v3 N) { return sub3(I, scale3(N, 2.0f * dot3(N, I))); }
/* A.3: refract */
static inline v3 refract3(v3 I, v3 N, float eta) {
    float d = dot3(N, I);
    float k = 1.0f - eta * (eta * (1.0f - d * d));
    if (k < 0.0f) return V3(0.0f, 0.0f, 0.0f);
    return sub3(scale3(I, eta), scale3(N, eta * d + sqrtf(k)));
}
static inline float fract1(float x) { return x - floorf(x); }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* pow(x, 5.0) (:222, :241).  llvmpipe evaluates exp2(5*log2(x)): NaN for x < 0 (GLSL leaves
 * pow undefined there; probed with gl_harness probe mode, tests/test_oracle_units.py, fixture tests/golden/probes.npz),
 * 0 for x = 0, and for x > 0 a value within 1.3e-6 rel of the true power.  The restatement
 * uses the exact-product form (<= 2 ulp from the true power), which the HIP kernel evaluates
 * with the same three multiplies, so the two agree bit for bit.
 * DIAGNOSTIC (orc_params.reserved0 bit 0, tests only): use mesa_pow5f below -- llvmpipe's own polynomial
 * pow, restated bit for bit -- instead; with it gColor equals the reference fixtures bit for bit as well,
 * which proves that the exact-product pow is the ONLY arithmetic difference left against the reference. */
/* (per THREAD, set from the render's own parameters at the start of every row: concurrent renders with different flags do
 *  not disturb each other -- ADVICE r2) */
static _Thread_local int g_mesa_pow = 0;
/* DIAGNOSTIC (orc_params.reserved0 bit 1, tests only): llvmpipe's LOOP LIMITER.  gallivm gives every shader ONE counter of
 * LP_MAX_TGSI_LOOP_ITERS = 65535 loop iterations, shared by all (inlined, nested) loops and never reset; every pass
 * through the end of any loop decrements it, and once it reaches zero every loop leaves at its next end.  A hang guard of
 * the software rasteriser, not a property of the shader -- but C5 at MAX_RAY_DEPTH >= 8 reaches it: 8 bounces x (1 + 8
 * lights x 4 samples) traversals x 257 passes of the 256-object loop = 67 848 > 65 535, so in the pixels whose path is
 * still alive at the 8th bounce the reference-on-llvmpipe stops adding lights (and, from depth 9 on, stops finding the
 * closest hit): its colour is LOWER than the shader's by percents with identical gPosition -- the residue of VERDICT r2
 * (8 of 14 400 px low-res, 7 of 59 904 at 8K; 0 px for every depth <= 7; C4's 64 objects stay below the limit).  With
 * this bit the restatement counts the same passes per pixel (llvmpipe counts per 8-invocation vector, whose loops have
 * lane-independent trip counts here) and leaves its loops the same way. */
static _Thread_local int g_lim = 0, g_budget = 0, g_lim_sss_lane = 0;
static inline int loop_end(void) { return g_lim && (--g_budget <= 0); }
/* a loop whose every pass through its end -- the pass that finds the condition false included, as in gallivm's
 * structured loops -- takes one tick of the limiter; `break` inside the body must be written LIM_BREAK */
#define LIM_FOR(decl, cond, inc) \
    for (int lim_stop_ = 0, lim_once_ = 1; lim_once_; lim_once_ = 0) for (decl; ((cond) || (loop_end(), 0)) && !lim_stop_; inc, lim_stop_ = loop_end())
#define LIM_BREAK { (void)loop_end(); break; }
static inline float mesa_pow5f(float x);
static inline float pow5(float x) {
    if (g_mesa_pow) return mesa_pow5f(x);
    if (x < 0.0f) return NAN;
    float x2 = x * x;
    return (x2 * x2) * x;
}

/* llvmpipe's log2 / exp2 (gallivm lp_build_log2_approx, lp_build_exp2) -- restated from the published algorithm:
 *   log2(x) = y*P(y*y) + exponent,  y = (m-1)/(m+1), m = mantissa in [1,2), P of degree 4;
 *   exp2(x) = 2^floor(x) * Q(x - floor(x)), Q of degree 5, x clamped to [-126.99999, 128];
 * polynomials by gallivm's even/odd Horner split (lp_build_polynomial), every multiply-add FUSED (llvm.fmuladd on
 * this FMA3 host).  The coefficient values are those of the installed Mesa 23.2.1 (read from the library's
 * constant tables, they are Mesa's published minimax fits).  pow(x,y) = exp2(log2(x)*y) and exp(x) =
 * exp2(x*fl(log2 e)) as NIR lowers them.  Pinned BITWISE against llvmpipe (tests/golden/trig.npz, 2^18 arguments:
 * log2, exp2, exp, pow(x,5) 100 % equal). */
static inline float mesa_poly(float x, const float *c, int n) {
    float x2 = x * x, even = 0.0f, odd = 0.0f;
    int he = 0, ho = 0;
    for (int i = n; i--;) {
        if (i % 2 == 0) { even = he ? fmaf(x2, even, c[i]) : c[i]; he = 1; }
        else            { odd = ho ? fmaf(x2, odd, c[i]) : c[i]; ho = 1; }
    }
    return ho ? fmaf(odd, x, even) : even;
}
static inline float mesa_log2f(float x) {
    static const float P[5] = {2.88539009343309178325f, 0.961791550404184197881f, 0.577440339438736392009f,
                               0.403343858251329912514f, 0.406718052498846252698f};
    uint32_t i;
    memcpy(&i, &x, 4);
    float logexp = (float)((int32_t)((i & 0x7f800000u) >> 23) - 127);
    uint32_t mi = (i & 0x007fffffu) | 0x3f800000u;
    float m;
    memcpy(&m, &mi, 4);
    float y = (m - 1.0f) / (m + 1.0f);
    float res = fmaf(y, mesa_poly(y * y, P, 5), logexp);
    if (x >= INFINITY) res = INFINITY;
    if (x == 0.0f) res = -INFINITY;
    if (!(x >= 0.0f)) res = NAN;      /* negative or NaN */
    return res;
}
static inline float mesa_exp2f(float x) {
    static const float Q[6] = {1.0f, 0.693153073200168932794f, 0.240153617044375388211f, 0.0558263180532956664775f,
                               0.00898934009049466391101f, 0.00187757667519147912699f};
    if (x != x) return x;
    x = fmaxf(-126.99999f, fminf(128.0f, x));
    float ip = floorf(x), fp = x - ip;
    uint32_t eb = (uint32_t)((int32_t)ip + 127) << 23;
    float e;
    memcpy(&e, &eb, 4);
    return e * mesa_poly(fp, Q, 6);
}
/* lp_build_pow: exp2(log2(x)*y), then 0 where x == 0 by an UNORDERED compare (true for NaN too: pow(NaN, 5) = 0) */
static inline float mesa_pow5f(float x) { return (x == 0.0f || x != x) ? 0.0f : mesa_exp2f(mesa_log2f(x) * 5.0f); }
static inline float mesa_expf(float x) { return mesa_exp2f(x * 1.44269504088896340736f); }

/* sin / cos / tan as the reference's GL evaluates them.  Mesa's GLSL front end defines tan(x) as
 * sin(x)/cos(x) and llvmpipe (gallivm lp_build_sin_or_cos, a port of the cephes / sse_mathfun single-precision
 * sincos) evaluates both with: |x| scaled by 4/pi, j = (int(y)+1) & ~1, a three-constant Cody-Waite reduction
 * and two degree-3 polynomials in z = r*r -- with FUSED multiply-adds exactly where gallivm emits llvm.fmuladd
 * (this host has FMA3; everything else is separate mul / add), the result clamped to [-1, 1] and NaN for a
 * non-finite argument.  Restated from that published algorithm and pinned BITWISE against llvmpipe itself
 * (tests/golden/trig.npz: 2^20 arguments incl. |x| up to 1e6, multiples of pi/2, denormals, inf/NaN: 100 %
 * equal for sin, cos and tan; tests/test_oracle_units.py).  Exact for |x| < 1.6e9 (beyond, int(y) overflows).
 * Used for tan(radians(fov)*0.5) (:209), cosineWeightedHemisphere's cos/sin(phi) (:296-298) and random()'s
 * sin (:274), so the camera rays, the bounce samples and the Russian-roulette decisions are the reference's
 * own, bit for bit.  The HIP side restates the same algorithm (rt_abi.cpp host, rt_kernels.hip device). */
static inline float mesa_sincosf(float a, int want_cos) {
    uint32_t ai;
    memcpy(&ai, &a, 4);
    uint32_t xi = ai & 0x7fffffffu;
    float x;
    memcpy(&x, &xi, 4);
    float y = x * 1.27323954473516f;                       /* 4/pi */
    int32_t j = (fabsf(y) < 2147483648.0f) ? (int32_t)y : INT32_MIN;   /* cvttps2dq */
    int32_t jadd = (int32_t)((uint32_t)j + 1u);
    int32_t jand = jadd & ~1;
    float y2 = (float)jand;
    int32_t e2 = want_cos ? (int32_t)((uint32_t)jand - 2u) : jand;
    uint32_t sign = want_cos ? (((uint32_t)(4 & ~e2)) << 29) : ((ai ^ ((uint32_t)jadd << 29)) & 0x80000000u);
    int use_sin_poly = (e2 & 2) == 0;
    float r = fmaf(y2, -0.78515625f, x);
    r = fmaf(y2, -2.4187564849853515625e-4f, r);
    r = fmaf(y2, -3.77489497744594108e-8f, r);
    float z = r * r;
    float c = fmaf(z, 2.443315711809948E-005f, -1.388731625493765E-003f);
    c = fmaf(c, z, 4.166664568298827E-002f);
    c = (c * z) * z;
    c = (c - z * 0.5f) + 1.0f;
    float s_ = fmaf(z, -1.9515295891E-4f, 8.3321608736E-3f);
    s_ = fmaf(s_, z, -1.6666654611E-1f);
    s_ = fmaf(s_ * z, r, r);
    float v = use_sin_poly ? s_ : c;
    uint32_t vi;
    memcpy(&vi, &v, 4);
    vi ^= sign;
    memcpy(&v, &vi, 4);
    v = fminf(fmaxf(v, -1.0f), 1.0f);
    if (!(fabsf(a) < INFINITY)) v = NAN;
    return v;
}
static inline float mesa_sinf(float a) { return mesa_sincosf(a, 0); }
static inline float mesa_cosf(float a) { return mesa_sincosf(a, 1); }
static inline float mesa_tanf(float a) { return mesa_sincosf(a, 0) / mesa_sincosf(a, 1); }

/* test hook (tests/test_oracle_units.py): out[3i..3i+2] = sin, cos, tan of in[i] */
void orc_mesa_trig(const float *in, float *out, int n) {
    for (int i = 0; i < n; i++) {
        out[3 * i] = mesa_sinf(in[i]);
        out[3 * i + 1] = mesa_cosf(in[i]);
        out[3 * i + 2] = mesa_tanf(in[i]);
    }
}
/* test hook: out[4i..4i+3] = log2, exp2, pow(x,5), exp of in[i] */
void orc_mesa_explog(const float *in, float *out, int n) {
    for (int i = 0; i < n; i++) {
        out[4 * i] = mesa_log2f(in[i]);
        out[4 * i + 1] = mesa_exp2f(in[i]);
        out[4 * i + 2] = mesa_pow5f(in[i]);
        out[4 * i + 3] = mesa_expf(in[i]);
    }
}

/* ------------------------------------------------------------------ scene records */
typedef struct { /* Material, raytracingCs.glsl:20-32; bytes 64..143 of Object */
    v3 albedo;
    float metallic, roughness, diffuseStrength, ior, transparency;
    float subsurfaceScatter;
    v3 subsurfaceColor;
    float scatterDistance;
} Mat;

typedef struct { /* Object, :34-42 (176-byte std430 stride) */
    int32_t type;
    v3 position;
    float radius;
    v3 normal;
    float size[2];
    Mat mat;
    v3 bmin, bmax;
} Obj;

typedef struct { /* Light, :44-58 (96-byte stride) */
    int32_t type;
    v3 position, direction, color;
    float intensity;
    float shadowSoftness;
    int32_t shadowType, pcfSamples;
    float lightSize, angularRadius;
} Lgt;

static float rdf(const uint8_t *p, int off) { float f; memcpy(&f, p + off, 4); return f; }
static int32_t rdi(const uint8_t *p, int off) { int32_t i; memcpy(&i, p + off, 4); return i; }
static v3 rd3(const uint8_t *p, int off) { return V3(rdf(p, off), rdf(p, off + 4), rdf(p, off + 8)); }

static void decode_object(const uint8_t *p, Obj *o) { /* offsets: SURVEY.md Appendix B */
    o->type = rdi(p, 0);
    o->position = rd3(p, 16);
    o->radius = rdf(p, 28);
    o->normal = rd3(p, 32);
    o->size[0] = rdf(p, 48);
    o->size[1] = rdf(p, 52);
    o->mat.albedo = rd3(p, 80);
    o->mat.metallic = rdf(p, 92);
    o->mat.roughness = rdf(p, 96);
    o->mat.diffuseStrength = rdf(p, 100);
    o->mat.ior = rdf(p, 104);
    o->mat.transparency = rdf(p, 108);
    o->mat.subsurfaceScatter = rdf(p, 116);
    o->mat.subsurfaceColor = rd3(p, 128);
    o->mat.scatterDistance = rdf(p, 140);
    o->bmin = rd3(p, 144);
    o->bmax = rd3(p, 160);
}

static void decode_light(const uint8_t *p, Lgt *l) {
    l->type = rdi(p, 0);
    l->position = rd3(p, 16);
    l->direction = rd3(p, 32);
    l->color = rd3(p, 48);
    l->intensity = rdf(p, 60);
    l->shadowSoftness = rdf(p, 72);
    l->shadowType = rdi(p, 76);
    l->pcfSamples = rdi(p, 80);
    l->lightSize = rdf(p, 84);
    l->angularRadius = rdf(p, 88);
}

typedef struct {
    const Obj *objs; int nObj;
    const Lgt *lts; int nLt;
    const orc_params *p;
    const uint8_t *noise; int noiseW, noiseH;
    const uint16_t *sky; int skySize;
    /* per-pixel state */
    uint32_t gidx, gidy;
    uint64_t rays;
} Ctx;

typedef struct { v3 origin, direction; } Ray; /* :13-18; energy/depth are dead (A.1#21) */

/* ------------------------------------------------------------------ half helpers */
static float half_to_float(uint16_t h) {
    uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023, u;
    if (e == 0) {
        if (m == 0) u = s;
        else {
            int sh = 0;
            while (!(m & 1024)) { m <<= 1; sh++; }
            m &= 1023;
            u = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}

/* A.3: imageStore to rgba16f rounds toward zero */
static uint16_t float_to_half_rtz(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    uint32_t s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) { /* inf / nan */
        if (a == 0x7f800000u) return (uint16_t)(s | 0x7c00u);
        return (uint16_t)(s | 0x7e00u | ((a >> 13) & 0x1ffu));
    }
    if (a >= 0x47800000u) return (uint16_t)(s | 0x7bffu); /* >= 65536: RTZ saturates to max finite */
    if (a >= 0x38800000u) return (uint16_t)(s | ((a - 0x38000000u) >> 13)); /* normal */
    if (a < 0x33800000u) return (uint16_t)s; /* < 2^-24 */
    uint32_t e = a >> 23, m = (a & 0x7fffffu) | 0x800000u;
    return (uint16_t)(s | (m >> (126 - e))); /* subnormal: truncate */
}

/* ------------------------------------------------------------------ textures */
/* texture(blueNoiseTex, uv): R8 unorm, NEAREST, REPEAT.
 * NULL texture = shipped behaviour: the sample reads 0. Returns .r (g=b=0). */
static float sample_noise(const Ctx *c) {
    if (!c->noise) return 0.0f;
    /* (gl_GlobalInvocationID.xy + frameCount) * noiseScale   :513, :359 */
    float u = (float)(uint32_t)(c->gidx + (uint32_t)c->p->frameCount) * c->p->noiseScale[0];
    float v = (float)(uint32_t)(c->gidy + (uint32_t)c->p->frameCount) * c->p->noiseScale[1];
    u = fract1(u); v = fract1(v);
    int ix = (int)floorf(u * (float)c->noiseW), iy = (int)floorf(v * (float)c->noiseH);
    if (ix >= c->noiseW) ix = c->noiseW - 1;
    if (iy >= c->noiseH) iy = c->noiseH - 1;
    if (ix < 0) ix = 0;
    if (iy < 0) iy = 0;
    return (float)c->noise[(size_t)iy * c->noiseW + ix] * (1.0f / 255.0f);
}

/* texture(samplerCube, d): LINEAR, CLAMP_TO_EDGE per face, non-seamless, RGB16F
 * (A.3 table row "texture(samplerCube, d)"). */
static v3 sample_cube(const uint16_t *sky, int size, v3 d) {
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
    int x0 = (int)fu, y0 = (int)fv, x1 = x0 + 1, y1 = y0 + 1;
    x0 = clampi(x0, 0, size - 1); x1 = clampi(x1, 0, size - 1);
    y0 = clampi(y0, 0, size - 1); y1 = clampi(y1, 0, size - 1);
    const uint16_t *f = sky + (size_t)face * size * size * 3;
    float out[3];
    for (int ch = 0; ch < 3; ch++) {
        float c00 = half_to_float(f[((size_t)y0 * size + x0) * 3 + ch]);
        float c10 = half_to_float(f[((size_t)y0 * size + x1) * 3 + ch]);
        float c01 = half_to_float(f[((size_t)y1 * size + x0) * 3 + ch]);
        float c11 = half_to_float(f[((size_t)y1 * size + x1) * 3 + ch]);
        float a = c00 + wu * (c10 - c00);
        float b = c01 + wu * (c11 - c01);
        out[ch] = a + wv * (b - a);
    }
    return V3(out[0], out[1], out[2]);
}

/* ------------------------------------------------------------------ intersections */
/* intersectAABB  :91-103.  min/max with a NaN operand return the other one (A.3). */
static int intersectAABB(const Ctx *c, const Ray *r, v3 bmin, v3 bmax) {
    v3 invDir = div3(splat3(1.0f), r->direction);
    v3 t0 = mul3(sub3(bmin, r->origin), invDir);
    v3 t1 = mul3(sub3(bmax, r->origin), invDir);
    v3 ts = V3(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z));
    v3 tl = V3(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y), fmaxf(t0.z, t1.z));
    float tMin = fmaxf(fmaxf(ts.x, ts.y), ts.z);
    float tMax = fminf(fminf(tl.x, tl.y), tl.z);
    return tMax >= tMin && tMin < c->p->maxRayDistance && tMax > 0.0f;
}

/* intersectSphere  :105-118 */
static int intersectSphere(const Ray *r, const Obj *o, float *t) {
    v3 oc = sub3(r->origin, o->position);
    float a = dot3(r->direction, r->direction);
    float b = 2.0f * dot3(oc, r->direction);
    float cc = dot3(oc, oc) - o->radius * o->radius;
    float disc = b * b - 4.0f * a * cc;
    if (disc < 0.0f) return 0;
    *t = (-b - sqrtf(disc)) / (2.0f * a);
    return *t > 0.0f;
}

/* intersectPlane  :120-153 (normal used unnormalised, A.1#8) */
static int intersectPlane(const Ray *r, const Obj *o, float *t) {
    float denom = dot3(o->normal, r->direction);
    if (fabsf(denom) > 1e-6f) {
        *t = dot3(sub3(o->position, r->origin), o->normal) / denom;
        if (*t < 0.0f) return 0;
        v3 hitPoint = add3(r->origin, scale3(r->direction, *t));
        v3 right, forward;
        if (fabsf(o->normal.y) > 0.9f) right = normalize3(cross3(o->normal, V3(0, 0, 1)));
        else right = normalize3(cross3(o->normal, V3(0, 1, 0)));
        forward = normalize3(cross3(right, o->normal));
        v3 lo = sub3(hitPoint, o->position);
        float x = dot3(lo, right), z = dot3(lo, forward);
        if (fabsf(x) > o->size[0] / 2.0f || fabsf(z) > o->size[1] / 2.0f) return 0;
        return 1;
    }
    return 0;
}

/* intersectObjects  :155-196.  hitMat/hitNormal are written only on improvement;
 * t always leaves as minT (:194). */
static int intersectObjects(Ctx *c, const Ray *r, Mat *hitMat, v3 *hitNormal, float *t) {
    float minT = c->p->maxRayDistance;
    int hit = 0;
    c->rays++;
    LIM_FOR(int i = 0, i < c->nObj, i++) {
        const Obj *o = &c->objs[i];
        if (!intersectAABB(c, r, o->bmin, o->bmax)) continue;
        float ct = 0.0f;
        int isHit = 0;
        if (o->type == 0) isHit = intersectSphere(r, o, &ct);
        else if (o->type == 1) isHit = intersectPlane(r, o, &ct);
        if (isHit && ct > 0.0f && ct < minT) {
            minT = ct;
            hit = 1;
            *hitMat = o->mat;
            if (o->type == 0)
                *hitNormal = normalize3(sub3(add3(r->origin, scale3(r->direction, ct)), o->position));
            else
                *hitNormal = o->normal;
        }
    }
    *t = minT;
    return hit;
}

/* ------------------------------------------------------------------ sampling helpers */
/* haltonSequence  :278-288 */
static float halton(int index, int base) {
    float result = 0.0f;
    float f = 1.0f / (float)base;
    int i = index;
    LIM_FOR(int k_ = 0, i > 0, k_++) {
        result += f * (float)(i % base);
        i = i / base;
        f = f / (float)base;
    }
    return result;
}

/* cosineWeightedHemisphere  :291-308 */
static v3 cosineWeightedHemisphere(float rx, float ry, v3 n) {
    float phi = 2.0f * PI_F * rx;
    float cosTheta = sqrtf(ry);
    float sinTheta = sqrtf(1.0f - ry);
    v3 h = V3(sinTheta * mesa_cosf(phi), cosTheta, sinTheta * mesa_sinf(phi));
    v3 tangent = normalize3(cross3(n, V3(0, 1, 1)));
    v3 bitangent = cross3(n, tangent);
    /* tangent = (n.y - n.z, -n.x, n.x) * rsq, so bitangent.x = n.y*t.z - n.z*t.y = n.y*t.z + n.z*t.z, which Mesa's NIR factors
     * into t.z * (n.y + n.z) (final NIR of the reference shader, LP_DEBUG=cs: `fadd %226, %227` then `fmul %2326, %2331`).
     * One rounding fewer; invisible while the hemisphere sample's sin(phi) is ~0 (frameCount 0: C2, C4, C5) and the cause
     * of C3's 20-of-32 400-pixel residue of round 2 (frameCount 7): with it gPosition equals the reference's bit for bit. */
    bitangent.x = tangent.z * (n.y + n.z);
    return normalize3(add3(add3(scale3(tangent, h.x), scale3(bitangent, h.z)), scale3(n, h.y)));
}

/* random  :273-275 (vec2 dot: a.y*b.y + a.x*b.x, same z->y->x order as vec3) */
static float random2(float sx, float sy) {
    float d = sy * 78.233f + sx * 12.9898f;
    return fract1(mesa_sinf(d) * 43758.5453123f);
}

/* fresnelSchlick  :220-223; pow(x,2.0) = x*x (A.3) */
static float fresnelSchlick(float cosTheta, float ior) {
    float q = (1.0f - ior) / (1.0f + ior);
    float r0 = q * q;
    return r0 + (1.0f - r0) * pow5(1.0f - cosTheta);
}

/* computePBR  :226-253 */
static v3 computePBR(const Mat *m, v3 N, v3 V, v3 L, v3 H, v3 radiance) {
    float alpha = m->roughness * m->roughness;
    float NdotH = fmaxf(dot3(N, H), 0.0f);
    /* x*(a2 - 1.0) + 1.0 is recognised by Mesa's NIR as lerp(1.0, a2, x) and evaluated as
     * (1.0 - x) + a2*x (probed bitwise: 100 % of 4096 samples; the as-written form matches 2 %) */
    float nh2 = NdotH * NdotH;
    float inner = (1.0f - nh2) + (alpha * alpha) * nh2;
    /* PI * pow(inner, 2.0): pow folds to inner*inner and NIR re-associates the constant
     * first, (PI*inner)*inner (probed bitwise, 100 % of 4096 samples) */
    float NDF = alpha * alpha / ((PI_F * inner) * inner);
    float rp1 = m->roughness + 1.0f;
    float k = (rp1 * rp1) / 8.0f;
    float NdotV = fmaxf(dot3(N, V), 0.0f), NdotL = fmaxf(dot3(N, L), 0.0f);
    float G = NdotV / (NdotV * (1.0f - k) + k);
    G *= NdotL / (NdotL * (1.0f - k) + k);
    v3 F0 = mix3_strict(splat3(0.04f), m->albedo, m->metallic);
    float p5 = pow5(1.0f - fmaxf(dot3(H, V), 0.0f));
    v3 F = add3(F0, scale3(sub3(splat3(1.0f), F0), p5));
    v3 numerator = scale3(F, NDF * G);
    float denominator = 4.0f * NdotV * NdotL;
    v3 specular = divs3(numerator, fmaxf(denominator, 0.001f));
    v3 kD = scale3(sub3(splat3(1.0f), F), 1.0f - m->metallic);
    v3 diffuse = divs3(mul3(kD, m->albedo), PI_F);
    return scale3(mul3(add3(diffuse, specular), radiance), NdotL);
}

/* ------------------------------------------------------------------ shadows */
/* pcfShadow  :342-397.  The directional-light filterSize of :352 is dead (A.1#11). */
static float pcfShadow(Ctx *c, v3 point, v3 normal, const Lgt *l, v3 lightDir, float lightDistance) {
    float shadow = 0.0f;
    v3 tangent = normalize3(cross3(lightDir, V3(0, 1, 0)));
    v3 bitangent = cross3(lightDir, tangent);
    float jr = sample_noise(c); /* .rg of an R8 texture = (r, 0) */
    LIM_FOR(int i = 0, i < l->pcfSamples, i++) {
        float filterSize = l->shadowSoftness * 0.005f;
        float rx = fract1(halton(i, 2) + jr);
        float ry = fract1(halton(i, 3) + 0.0f);
        v3 jd = add3(add3(lightDir, scale3(scale3(tangent, rx), filterSize)),
                     scale3(scale3(bitangent, ry), filterSize));
        if (l->type != 1) jd = normalize3(jd);
        Ray sr;
        sr.origin = add3(point, scale3(normal, 0.001f));
        sr.direction = jd;
        Mat tm; v3 tn; float t;
        int occ = intersectObjects(c, &sr, &tm, &tn, &t);
        if (l->type == 0 || l->type == 2) occ = occ && (t < lightDistance);
        shadow += occ ? 0.0f : 1.0f;
    }
    return shadow / (float)l->pcfSamples;
}

/* pcssShadow  :400-440 */
static float pcssShadow(Ctx *c, v3 point, v3 normal, const Lgt *l, v3 lightDir, float lightDistance) {
    int blockerCount = 0;
    float searchSize = l->lightSize * 0.1f;
    LIM_FOR(int i = 0, i < 16, i++) {
        float rr = halton(i, 3) * 2.0f - 1.0f;
        v3 sd = add3(add3(lightDir, splat3(rr * searchSize)), splat3(rr * searchSize));
        Ray sr;
        sr.origin = add3(point, scale3(normal, 0.001f));
        sr.direction = normalize3(sd);
        Mat tm; v3 tn; float t;
        int occ = intersectObjects(c, &sr, &tm, &tn, &t);
        if (l->type != 1) occ = occ && (t < lightDistance);
        if (occ) blockerCount++;
    }
    if (blockerCount == 0) return 1.0f;
    return pcfShadow(c, point, normal, l, lightDir, lightDistance);
}

/* calculateShadow  :442-455 */
static float calculateShadow(Ctx *c, v3 point, v3 normal, v3 lightDir, float lightDistance, const Lgt *l) {
    if (l->shadowType == 0) return 1.0f;
    float shadow = 0.0f;
    if (l->shadowType == 1) shadow = pcfShadow(c, point, normal, l, lightDir, lightDistance);
    else if (l->shadowType == 2) shadow = pcssShadow(c, point, normal, l, lightDir, lightDistance);
    return shadow;
}

/* computeSubsurfaceScattering  :316-339 */
static v3 computeSSS(Ctx *c, v3 P, v3 N, const Mat *m) {
    v3 sss = V3(0, 0, 0);
    LIM_FOR(int i = 0, i < 4, i++) {
        float rx = (float)i / 4.0f, ry = halton(i, 2); /* hammersley(i,4) :311-313 */
        Ray r;
        r.origin = add3(P, scale3(N, 0.001f));
        r.direction = cosineWeightedHemisphere(rx, ry, N);
        Mat tm; v3 tn; float t;
        if (intersectObjects(c, &r, &tm, &tn, &t)) {
            float att = mesa_expf(-t / m->scatterDistance);
            sss = add3(sss, scale3(tm.albedo, att));
        }
    }
    return divs3(scale3(mul3(sss, m->subsurfaceColor), m->subsurfaceScatter), 4.0f);
}

/* computeLighting  :457-507 */
static v3 computeLighting(Ctx *c, v3 P, v3 N, const Mat *m, v3 V) {
    v3 Lo = V3(0, 0, 0);
    LIM_FOR(int i = 0, i < c->nLt, i++) {
        const Lgt *l = &c->lts[i];
        v3 lightDir = V3(0, 0, 0);
        float attenuation = 1.0f, lightDistance = 0.0f;
        if (l->type == 0) {
            lightDir = sub3(l->position, P);
            lightDistance = length3(lightDir);
            attenuation = 1.0f / (1.0f + 0.1f * lightDistance + 0.01f * lightDistance * lightDistance);
            lightDir = normalize3(lightDir);
        } else if (l->type == 1) {
            lightDir = normalize3(neg3(l->direction));
            lightDistance = 1e6f;
        } else if (l->type == 2) {
            lightDir = sub3(l->position, P);
            /* lightDistance*lightDistance = sqrt(q)*sqrt(q): Mesa's NIR folds it to |q| with
             * q = dot(lightDir, lightDir), i.e. no sqrt rounding (probed in-shader, 100 %) */
            float q = dot3(lightDir, lightDir);
            lightDistance = length3(lightDir);
            lightDir = normalize3(lightDir);
            attenuation = 1.0f / fabsf(q);
            v3 ln = normalize3(l->direction);
            float lc = fmaxf(dot3(lightDir, ln), 0.0f);
            attenuation *= lc;
        }
        float shadowFactor = calculateShadow(c, P, N, lightDir, lightDistance, l);
        v3 L = normalize3(lightDir);
        v3 H = normalize3(add3(V, L));
        v3 radiance = scale3(scale3(l->color, attenuation), l->intensity);
        Lo = add3(Lo, scale3(computePBR(m, N, V, L, H, radiance), shadowFactor));
    }
    if (m->subsurfaceScatter > 0.0f) Lo = add3(Lo, computeSSS(c, P, N, m));
    else if (g_lim && g_lim_sss_lane) {
        /* (diagnostic bit 2, with bit 1) llvmpipe's limiter counts per VECTOR of 8 invocations: when another lane of the vector
         * shades a subsurface material, the whole vector passes through computeSubsurfaceScattering's loops -- four probe
         * traversals with their Halton loops -- and this pixel's budget shrinks by the same passes */
        for (int i = 0; i < 4; i++) {
            for (int k = i; k > 0; k /= 2) (void)loop_end();
            (void)loop_end();                                        /* haltonSequence(i, 2): digits + the leaving pass */
            for (int k = 0; k <= c->nObj; k++) (void)loop_end();     /* intersectObjects */
            (void)loop_end();                                        /* the probe loop's own end */
        }
        (void)loop_end();
    }
    return Lo;
}

/* calculateRefraction  :256-270 (energy update is a dead store, A.1#21) */
static v3 calculateRefraction(const Ray *r, v3 N, const Mat *m) {
    int entering = dot3(r->direction, N) < 0.0f;
    float eta = entering ? (1.0f / m->ior) : m->ior;
    v3 normal = entering ? N : neg3(N);
    v3 rd = refract3(normalize3(r->direction), normal, eta);
    if (dot3(rd, rd) < 0.001f) rd = reflect3(r->direction, normal);
    return rd;
}

/* generateCameraRay  :198-217 */
static void generateCameraRay(const Ctx *c, Ray *ray, float jx, float jy) {
    const orc_params *p = c->p;
    float ux = (((float)(int)c->gidx + 0.5f) + jx) / (float)p->width;
    float uy = (((float)(int)c->gidy + 0.5f) + jy) / (float)p->height;
    ux = ux * 2.0f - 1.0f;
    uy = uy * 2.0f - 1.0f;
    float aspect = (float)p->width / (float)p->height;
    float tanFov = mesa_tanf((p->fovDeg * 0.017453292519943295f) * 0.5f);   /* radians(x) = x*fl(pi/180), A.3 */
    ux *= aspect * tanFov * p->focalLength;
    uy *= tanFov * p->focalLength;
    v3 cd = V3(p->camDir[0], p->camDir[1], p->camDir[2]);
    v3 cr = V3(p->camRight[0], p->camRight[1], p->camRight[2]);
    v3 cu = V3(p->camUp[0], p->camUp[1], p->camUp[2]);
    ray->origin = V3(p->camPos[0], p->camPos[1], p->camPos[2]);
    ray->direction = normalize3(add3(add3(cd, scale3(cr, ux)), scale3(cu, uy)));
}

/* main  :509-584 for one pixel */
static void shade_pixel(Ctx *c, float *color, float *pos, uint16_t *nrm) {
    const orc_params *p = c->p;
    float nz = sample_noise(c);
    float jx = nz * 2.0f - 1.0f, jy = 0.0f * 2.0f - 1.0f; /* .y of an R8 sample is 0 (A.1#3) */
    Ray ray;
    generateCameraRay(c, &ray, jx, jy);
    v3 finalColor = V3(0, 0, 0), throughput = V3(1, 1, 1);
    v3 P = V3(0, 0, 0), V, N = V3(0, 0, 0); /* undefined locals read as zero (A.3) */
    g_budget = 65535;        /* LP_MAX_TGSI_LOOP_ITERS, per invocation (only read under the diagnostic bit) */
    LIM_FOR(int depth = 0, depth < p->maxRayDepth, ++depth) {
        Mat mat; float t;
        memset(&mat, 0, sizeof mat);
        if (!intersectObjects(c, &ray, &mat, &N, &t)) {
            if (p->useSkybox && c->sky)
                finalColor = add3(finalColor, mul3(throughput, sample_cube(c->sky, c->skySize, ray.direction)));
            /* else: `finalColor += throughput * vec3(0.0)` (:532).  Mesa folds x*0.0 to 0.0
             * (inexact algebra), so a NaN/inf throughput does NOT poison the colour on a miss:
             * pinned by the nan fixture (tests/golden/nan.npz). */
            LIM_BREAK
        }
        P = add3(ray.origin, scale3(ray.direction, t));
        V = normalize3(neg3(ray.direction));
        v3 Lo = computeLighting(c, P, N, &mat, V);
        finalColor = add3(finalColor, mul3(throughput, Lo));
        if (depth > 2) { /* Russian roulette :544-549 */
            float dw = length3(mat.albedo) * mat.diffuseStrength;
            float cp = fminf(fmaxf(throughput.x, fmaxf(throughput.y, throughput.z)) * 0.95f + dw, 0.99f);
            float rnd = random2((float)(c->gidx + (uint32_t)depth), (float)(c->gidy + (uint32_t)depth));
            if (rnd > cp) LIM_BREAK
            throughput = divs3(throughput, cp);
        }
        float F = fresnelSchlick(fmaxf(dot3(V, N), 0.0f), mat.ior);
        if (mat.diffuseStrength > 0.0f) { /* :555-567 */
            int hi = depth * 64 + p->frameCount;
            float rx = (float)hi / 64.0f, ry = halton(hi, 2);
            v3 sd = reflect3(ray.direction, N);
            v3 dd = cosineWeightedHemisphere(rx, ry, N);
            v3 md = mix3_fast(sd, dd, mat.roughness);
            ray.direction = normalize3(md);
            ray.origin = add3(P, scale3(N, 0.001f));
            throughput = mul3(throughput, scale3(mat.albedo, mat.diffuseStrength));
        } else if (mat.transparency > 0.0f) { /* :568-571 */
            ray.direction = calculateRefraction(&ray, N, &mat);
            ray.origin = sub3(P, scale3(N, 0.001f));
            throughput = mul3(throughput, scale3(scale3(mat.albedo, 1.0f - F), mat.transparency));
        } else { /* :572-576 */
            ray.direction = reflect3(ray.direction, N);
            ray.origin = add3(P, scale3(N, 0.001f));
            throughput = mul3(throughput, scale3(mat.albedo, F));
        }
    }
    color[0] = finalColor.x; color[1] = finalColor.y; color[2] = finalColor.z; color[3] = 1.0f;
    pos[0] = P.x; pos[1] = P.y; pos[2] = P.z; pos[3] = 1.0f;
    nrm[0] = float_to_half_rtz(N.x); nrm[1] = float_to_half_rtz(N.y);
    nrm[2] = float_to_half_rtz(N.z); nrm[3] = 0x3c00u;
}

int orc_render(const void *objects, int nObj, const void *lights, int nLt, const orc_params *p,
               const uint8_t *noise, int noiseW, int noiseH, const uint16_t *sky, int skySize,
               float *gColor, float *gPosition, uint16_t *gNormal, uint64_t *rayCount, int nthreads) {
    if (!p || nObj < 0 || nLt < 0 || p->regionW < 0 || p->regionH < 0) return -1;
    if (p->stripRows <= 0) return -1;
    if (p->stripCycleRows > 0) {
        if (p->stripOffsetRows < 0 || p->stripOffsetRows + p->stripRows > p->stripCycleRows) return -1;
    } else if (p->stripCount <= 0 || p->stripIndex < 0 || p->stripIndex >= p->stripCount) return -1;
    const int cycleRows = p->stripCycleRows > 0 ? p->stripCycleRows : p->stripRows * p->stripCount;
    const int offsetRows = p->stripCycleRows > 0 ? p->stripOffsetRows : p->stripIndex * p->stripRows;
    Obj *objs = malloc(sizeof(Obj) * (size_t)(nObj > 0 ? nObj : 1));
    Lgt *lts = malloc(sizeof(Lgt) * (size_t)(nLt > 0 ? nLt : 1));
    for (int i = 0; i < nObj; i++) decode_object((const uint8_t *)objects + (size_t)i * 176, &objs[i]);
    for (int i = 0; i < nLt; i++) decode_light((const uint8_t *)lights + (size_t)i * 96, &lts[i]);
    uint64_t total = 0;
    const int diag_pow = p->reserved0 & 1, diag_lim = (p->reserved0 >> 1) & 1, diag_sss = (p->reserved0 >> 2) & 1;      /* diagnostics (see pow5, loop_end) */
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int j = 0; j < p->regionH; j++) {
        Ctx c;
        g_mesa_pow = diag_pow;
        g_lim = diag_lim;
        g_lim_sss_lane = diag_sss;
        c.objs = objs; c.nObj = nObj; c.lts = lts; c.nLt = nLt; c.p = p;
        c.noise = noise; c.noiseW = noiseW; c.noiseH = noiseH;
        c.sky = sky; c.skySize = skySize; c.rays = 0;
        int ly = p->y0 + j;
        int gy = (ly / p->stripRows) * cycleRows + offsetRows + ly % p->stripRows;
        for (int i = 0; i < p->regionW; i++) {
            int gx = p->x0 + i;
            size_t o = ((size_t)j * p->regionW + i) * 4;
            if (gx >= p->width || gy >= p->height) { /* outside the image: imageStore discarded */
                memset(gColor + o, 0, 16); memset(gPosition + o, 0, 16); memset(gNormal + o, 0, 8);
                continue;
            }
            c.gidx = (uint32_t)gx; c.gidy = (uint32_t)gy;
            shade_pixel(&c, gColor + o, gPosition + o, gNormal + o);
        }
        total += c.rays;
    }
    if (rayCount) *rayCount = total;
    free(objs); free(lts);
    return 0;
}

/* GenerateAABBForObject  /root/reference/src/SceneIO.h:75-104 (glm fp32 semantics:
 * glm::normalize(v) = v * inversesqrt(dot(v,v)), dot = x*x + y*y + z*z). */
static void wr3(uint8_t *p, int off, v3 v) { memcpy(p + off, &v.x, 4); memcpy(p + off + 4, &v.y, 4); memcpy(p + off + 8, &v.z, 4); }
static v3 glm_normalize(v3 v) {
    float d = v.x * v.x + v.y * v.y + v.z * v.z;
    return scale3(v, 1.0f / sqrtf(d));
}
void orc_generate_aabb(void *objects, int n) {
    for (int i = 0; i < n; i++) {
        uint8_t *p = (uint8_t *)objects + (size_t)i * 176;
        int32_t type = rdi(p, 0);
        v3 pos = rd3(p, 16), nrm = rd3(p, 32);
        float radius = rdf(p, 28), sx = rdf(p, 48), sy = rdf(p, 52);
        if (type == 0) {
            wr3(p, 144, sub3(pos, splat3(radius)));
            wr3(p, 160, add3(pos, splat3(radius)));
        } else if (type == 1) {
            v3 right, forward;
            if (fabsf(nrm.y) > 0.9f) { right = V3(1, 0, 0); forward = V3(0, 0, 1); }
            else {
                right = glm_normalize(cross3(nrm, V3(0, 1, 0)));
                forward = glm_normalize(cross3(right, nrm));
            }
            v3 hx = scale3(right, sx / 2.0f), hy = scale3(forward, sy / 2.0f);
            v3 mn = sub3(sub3(pos, hx), hy), mx = add3(add3(pos, hx), hy);
            mn = add3(mn, scale3(nrm, 0.01f));
            mx = add3(mx, scale3(nrm, 0.01f));
            wr3(p, 144, mn);
            wr3(p, 160, mx);
        }
    }
}

void orc_halton(const int32_t *index, const int32_t *base, float *out, int n) {
    for (int i = 0; i < n; i++) out[i] = halton(index[i], base[i]);
}
void orc_float_to_half_rtz(const float *in, uint16_t *out, int n) {
    for (int i = 0; i < n; i++) out[i] = float_to_half_rtz(in[i]);
}
void orc_sample_cube(const uint16_t *sky, int skySize, const float *dirs, float *rgb, int n) {
    for (int i = 0; i < n; i++) {
        v3 c = sample_cube(sky, skySize, V3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
    }
}
