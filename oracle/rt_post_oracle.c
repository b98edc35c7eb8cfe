/*
 * rt_post_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the reference's TAA resolve
 * fragment shader (/root/reference/shader/taaFs.glsl:13-53) as driven by
 * /root/reference/src/ForwardShadingPipeline.cpp:231-260 (texture set-up :57-65, :90-107,
 * :115-126; full-screen quad global.cpp:13-39).  Parity status: PINNED to the shader itself run on
 * llvmpipe through gl_harness' postfx mode (fixture tests/golden/taa.npz), to 1e-5 relative --
 * the rasteriser's interpolated TexCoords and llvmpipe's bilinear weights are reproduced to fp32
 * rounding, not bit for bit.
 *
 * Texture semantics restated:
 *   uCurrentFrame  rgba32f, LINEAR, REPEAT          (outputTex, ForwardShadingPipeline.cpp:57-65)
 *   uHistory       rgba32f, LINEAR, CLAMP_TO_EDGE   (historyTex, :90-100)
 *   gNormal        rgba16f, NEAREST, REPEAT         (gNormalTex, :121-125)
 *   texelFetch outside the image returns 0 (robust buffer access on the reference's GL; pinned by
 *   the fixture's border pixels).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline int wrapi(int i, int n) { int r = i % n; return r < 0 ? r + n : r; }
static inline int clampi2(int i, int n) { return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); }

/* bilinear fetch of an rgba32f image at normalised (s,t): texel centres at (i+0.5)/W */
static void bilinear(const float *img, int W, int H, float s, float t, int clampEdge, float out[3]) {
    float x = s * (float)W - 0.5f, y = t * (float)H - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float wx = x - fx, wy = y - fy;
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    if (clampEdge) { x0 = clampi2(x0, W); x1 = clampi2(x1, W); y0 = clampi2(y0, H); y1 = clampi2(y1, H); }
    else { x0 = wrapi(x0, W); x1 = wrapi(x1, W); y0 = wrapi(y0, H); y1 = wrapi(y1, H); }
    for (int c = 0; c < 3; c++) {
        float c00 = img[((size_t)y0 * W + x0) * 4 + c], c10 = img[((size_t)y0 * W + x1) * 4 + c];
        float c01 = img[((size_t)y1 * W + x0) * 4 + c], c11 = img[((size_t)y1 * W + x1) * 4 + c];
        float a = c00 + wx * (c10 - c00);
        float b = c01 + wx * (c11 - c01);
        out[c] = a + wy * (b - a);
    }
}

/* current/history: W*H*4 floats; normal: W*H*4 floats (the rgba16f values widened exactly);
 * out: W*H*4 floats.  Row 0 = bottom row. */
void orc_taa_resolve(const float *current, const float *history, const float *normal, int W, int H,
                     float blendFactor, float jitterX, float jitterY, float *out) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < H; j++) {
        for (int i = 0; i < W; i++) {
            /* TexCoords of the full-screen quad at the pixel centre */
            float u = ((float)i + 0.5f) / (float)W, v = ((float)j + 0.5f) / (float)H;
            float ju = u + jitterX, jv = v + jitterY;                       /* taaFs.glsl:23 */
            float cur[3], his[3];
            bilinear(current, W, H, ju, jv, 0, cur);                        /* :24 */
            bilinear(history, W, H, u, v, 1, his);                          /* :27 */
            float mn[3] = {cur[0], cur[1], cur[2]}, mx[3] = {cur[0], cur[1], cur[2]};
            for (int dx = -1; dx <= 1; dx++)                                /* :30-37 */
                for (int dy = -1; dy <= 1; dy++) {
                    int x = i + dx, y = j + dy;
                    for (int c = 0; c < 3; c++) {
                        float nb = (x < 0 || y < 0 || x >= W || y >= H) ? 0.0f : current[((size_t)y * W + x) * 4 + c];
                        mn[c] = fminf(mn[c], nb);
                        mx[c] = fmaxf(mx[c], nb);
                    }
                }
            /* :40-45 normals, NEAREST/REPEAT */
            int px = wrapi((int)floorf(u * (float)W), W), py = wrapi((int)floorf(v * (float)H), H);
            int cx = wrapi((int)floorf(ju * (float)W), W), cy = wrapi((int)floorf(jv * (float)H), H);
            const float *pn = normal + ((size_t)py * W + px) * 4, *cn = normal + ((size_t)cy * W + cx) * 4;
            float d = (pn[2] * cn[2] + pn[1] * cn[1]) + pn[0] * cn[0];
            float bf = 0.0f;
            if (d < 0.9f) bf = blendFactor * 0.2f;
            float *o = out + ((size_t)j * W + i) * 4;
            for (int c = 0; c < 3; c++) {
                /* clipAABB :13-19 */
                float center = 0.5f * (mx[c] + mn[c]);
                float extents = 0.5f * (mx[c] - mn[c]);
                float clip = his[c] - center;
                clip = fminf(fmaxf(clip, -extents), extents);
                float h = center + clip;
                o[c] = h + bf * (cur[c] - h);                               /* mix(history, current, blendFactor) :51 */
            }
            o[3] = 1.0f;
        }
    }
}

/* host jitter of ForwardShadingPipeline.cpp:241-242 with global.cpp:41-51's haltonSequence */
static float halton_host(int index, int base) {
    float result = 0.0f, f = 1.0f / (float)base;
    int i = index;
    while (i > 0) { result += f * (float)(i % base); i = (int)floorf((float)(i / base)); f /= (float)base; }
    return result;
}
void orc_taa_jitter(int frameCount, int W, int H, float *jx, float *jy) {
    *jx = halton_host(frameCount % 8, 2) * 0.5f / (float)W;
    *jy = halton_host(frameCount % 8, 3) * 0.5f / (float)H;
}

/* ---------------------------------------------------------------------------------------------
 * Bloom: /root/reference/shader/brightness_extractFS.glsl, gaussian_blurFs.glsl,
 * bloom_combineFs.glsl as driven by /root/reference/src/ForwardShadingPipeline.cpp:189-228
 * (textures :67-88: two rgba16f ping-pong targets, LINEAR, CLAMP_TO_EDGE; scene = outputTex rgba32f,
 * LINEAR, REPEAT).  All taps fall on texel centres, so LINEAR filtering returns the texel (up to the
 * rasteriser's TexCoords ulps, see the TAA note); render-target writes round fp32 -> fp16 toward zero
 * (pinned by tests/golden/bloom.npz).
 * ------------------------------------------------------------------------------------------- */
/* render-target write fp32 -> fp16: round toward zero, like imageStore */
static uint16_t f2h_rt(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    uint32_t s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(s | (a == 0x7f800000u ? 0x7c00u : (0x7e00u | ((a >> 13) & 0x1ffu))));
    if (a >= 0x47800000u) return (uint16_t)(s | 0x7bffu);
    if (a >= 0x38800000u) return (uint16_t)(s | ((a - 0x38000000u) >> 13));
    if (a < 0x33800000u) return (uint16_t)s;
    uint32_t e = a >> 23, m = (a & 0x7fffffu) | 0x800000u;
    return (uint16_t)(s | (m >> (126u - e)));
}
static float h2f(uint16_t h) {
    uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023, u;
    if (e == 0) {
        if (m == 0) u = s;
        else { int sh = 0; while (!(m & 1024)) { m <<= 1; sh++; } m &= 1023; u = s | ((uint32_t)(113 - sh) << 23) | (m << 13); }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}

/* brightness extract (brightness_extractFS.glsl:11-19): scene rgba32f -> rgba16f bits */
void orc_bloom_extract(const float *scene, int W, int H, float threshold, uint16_t *out) {
#pragma omp parallel for schedule(static)
    for (int k = 0; k < W * H; k++) {
        const float *c = scene + (size_t)k * 4;
        float brightness = (c[2] * 0.0722f + c[1] * 0.7152f) + c[0] * 0.2126f;
        uint16_t *o = out + (size_t)k * 4;
        if (brightness > threshold) { o[0] = f2h_rt(c[0]); o[1] = f2h_rt(c[1]); o[2] = f2h_rt(c[2]); }
        else { o[0] = o[1] = o[2] = 0; }
        o[3] = 0x3c00u;
    }
}

/* one separable 9-tap pass (gaussian_blurFs.glsl:8-26), CLAMP_TO_EDGE, rgba16f in/out */
void orc_bloom_blur(const uint16_t *in, int W, int H, int horizontal, uint16_t *out) {
    static const float w[5] = {0.227027f, 0.1945946f, 0.1216216f, 0.054054f, 0.016216f};
#pragma omp parallel for schedule(static)
    for (int j = 0; j < H; j++)
        for (int i = 0; i < W; i++) {
            float r[3];
            const uint16_t *c = in + ((size_t)j * W + i) * 4;
            for (int ch = 0; ch < 3; ch++) r[ch] = h2f(c[ch]) * w[0];
            for (int t = 1; t < 5; t++) {
                int xp = horizontal ? clampi2(i + t, W) : i, yp = horizontal ? j : clampi2(j + t, H);
                int xm = horizontal ? clampi2(i - t, W) : i, ym = horizontal ? j : clampi2(j - t, H);
                const uint16_t *p = in + ((size_t)yp * W + xp) * 4, *m = in + ((size_t)ym * W + xm) * 4;
                for (int ch = 0; ch < 3; ch++) { r[ch] += h2f(p[ch]) * w[t]; r[ch] += h2f(m[ch]) * w[t]; }
            }
            uint16_t *o = out + ((size_t)j * W + i) * 4;
            o[0] = f2h_rt(r[0]); o[1] = f2h_rt(r[1]); o[2] = f2h_rt(r[2]); o[3] = 0x3c00u;
        }
}

/* combine (bloom_combineFs.glsl:10-14): scene + bloom * strength, fp32 out */
void orc_bloom_combine(const float *scene, const uint16_t *bloom, int W, int H, float strength, float *out) {
#pragma omp parallel for schedule(static)
    for (int k = 0; k < W * H; k++) {
        for (int ch = 0; ch < 3; ch++) out[(size_t)k * 4 + ch] = scene[(size_t)k * 4 + ch] + h2f(bloom[(size_t)k * 4 + ch]) * strength;
        out[(size_t)k * 4 + 3] = 1.0f;
    }
}

/* ---------------------------------------------------------------------------------------------
 * SSAO: /root/reference/shader/ssaoFs.glsl:16-46 and ssao_blurFs.glsl:11-29 as driven by
 * /root/reference/src/AO.cpp:86-117 (kernel samples and the 4x4 rotation texture: AO.cpp:23-51; gPosition
 * rgba32f / gNormal rgba16f, NEAREST, default wrap = REPEAT: ForwardShadingPipeline.cpp:115-126).
 * Upstream the pass is dead (its FBOs have no attachments and nothing samples the result); the value
 * restated here is the float the fragment shader writes.  llvmpipe lowering pinned by probes
 * (tests/golden/make_golden.py): M*v = ((c0*x + c1*y) + c2*z) + c3*w; projection*view*p is evaluated as
 * projection*(view*p); dot = (z*z + y*y) + x*x; smoothstep(0,1,x) = t*(t*(3 - 2t)), t = clamp(x,0,1);
 * NEAREST/REPEAT texel = ifloor(u*size) & (size-1) for power-of-two sizes, itrunc(min(fract(u), 1-ulp)*size)
 * otherwise.
 * ------------------------------------------------------------------------------------------- */
static int nearest_repeat(float u, int size) {
    if ((size & (size - 1)) == 0) return ((int)floorf(u * (float)size)) & (size - 1);
    float fr = u - floorf(u);
    if (!(fr < 1.0f)) fr = 0.99999994f;          /* lp_build_fract_safe */
    if (!(fr >= 0.0f)) fr = 0.0f;                /* NaN coordinate: fract clamps to 0 */
    return (int)(fr * (float)size);
}
static void nrm3(const float v[3], float o[3]) {
    float d = (v[2] * v[2] + v[1] * v[1]) + v[0] * v[0];
    float r = 1.0f / sqrtf(d);
    o[0] = v[0] * r; o[1] = v[1] * r; o[2] = v[2] * r;
}
static void mat4_vec(const float *m, const float v[4], float o[4]) {   /* column-major, w = 1 keeps c3 exact */
    for (int r = 0; r < 4; r++) o[r] = ((m[r] * v[0] + m[4 + r] * v[1]) + m[8 + r] * v[2]) + m[12 + r] * v[3];
}

/* position: W*H*4 floats, normal: W*H*4 floats (rgba16f widened), noise: nW*nH*4 floats, samples: 64*3,
 * projection / view: 16 floats column-major, out: W*H floats.  Row 0 = bottom row. */
void orc_ssao(const float *position, const float *normal, int W, int H, const float *noise, int nW, int nH,
              const float *samples, const float *projection, const float *view, float *out) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < H; j++) {
        for (int i = 0; i < W; i++) {
            const float u = ((float)i + 0.5f) / (float)W, v = ((float)j + 0.5f) / (float)H;
            const float *fp = position + ((size_t)nearest_repeat(v, H) * W + nearest_repeat(u, W)) * 4;    /* :18 */
            float n[3], rv[3];
            nrm3(normal + ((size_t)nearest_repeat(v, H) * W + nearest_repeat(u, W)) * 4, n);             /* :19 */
            const float nu = u * 200.0f, nv = v * 200.0f;                                                   /* :14,:20 */
            nrm3(noise + ((size_t)nearest_repeat(nv, nH) * nW + nearest_repeat(nu, nW)) * 4, rv);
            /* :23-25 */
            float d = (rv[2] * n[2] + rv[1] * n[1]) + rv[0] * n[0];
            float t0[3] = {rv[0] - n[0] * d, rv[1] - n[1] * d, rv[2] - n[2] * d}, t[3], b[3];
            nrm3(t0, t);
            b[0] = n[1] * t[2] - t[1] * n[2];
            b[1] = n[2] * t[0] - t[2] * n[0];
            b[2] = n[0] * t[1] - t[0] * n[1];
            float occlusion = 0.0f;
            for (int k = 0; k < 64; k++) {                                                                  /* :29-44 */
                const float *s = samples + 3 * k;
                float sp[4];
                for (int c = 0; c < 3; c++) sp[c] = (t[c] * s[0] + b[c] * s[1]) + n[c] * s[2];              /* TBN * samples[i] */
                for (int c = 0; c < 3; c++) sp[c] = fp[c] + sp[c] * 0.5f;                                   /* :32 */
                sp[3] = 1.0f;
                float vw[4], off[4];
                mat4_vec(view, sp, vw);
                mat4_vec(projection, vw, off);                                                              /* :36 */
                float ox = off[0] / off[3], oy = off[1] / off[3];                                           /* :37 */
                ox = ox * 0.5f + 0.5f;                                                                      /* :38 */
                oy = oy * 0.5f + 0.5f;
                const float sampleDepth = position[((size_t)nearest_repeat(oy, H) * W + nearest_repeat(ox, W)) * 4 + 2];   /* :41 */
                float x = 0.5f / fabsf(fp[2] - sampleDepth);                                                /* :44 */
                float tt = fminf(fmaxf(x, 0.0f), 1.0f);
                float rangeCheck = tt * (tt * (3.0f - 2.0f * tt));
                occlusion += (sampleDepth >= sp[2] + 0.025f ? 1.0f : 0.0f) * rangeCheck;                    /* :45 */
            }
            out[(size_t)j * W + i] = 1.0f - occlusion / 64.0f;                                              /* :47 */
        }
    }
}

/* ssao_blurFs.glsl:11-29 (one direction per pass; ssaoColorBuffer is NEAREST, default wrap REPEAT) */
void orc_ssao_blur(const float *in, int W, int H, int horizontal, float *out) {
    const float wgt[5] = {0.227027f, 0.1945946f, 0.1216216f, 0.054054f, 0.016216f};
    const float tx = 1.0f / (float)W, ty = 1.0f / (float)H;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < H; j++)
        for (int i = 0; i < W; i++) {
            const float u = ((float)i + 0.5f) / (float)W, v = ((float)j + 0.5f) / (float)H;
            float r = in[(size_t)nearest_repeat(v, H) * W + nearest_repeat(u, W)] * wgt[0];
            for (int k = 1; k < 5; k++) {
                const float du = horizontal ? tx * (float)k : 0.0f, dv = horizontal ? 0.0f : ty * (float)k;
                r += in[(size_t)nearest_repeat(v + dv, H) * W + nearest_repeat(u + du, W)] * wgt[k];
                r += in[(size_t)nearest_repeat(v - dv, H) * W + nearest_repeat(u - du, W)] * wgt[k];
            }
            out[(size_t)j * W + i] = r;
        }
}

/* ---------------------------------------------------------------------------------------------
 * Equirectangular -> cubemap: ConvertHDRToCubemap, /root/reference/src/TextureLoader.cpp:118-194,
 * with shader/skyboxVs.glsl + skyboxFs.glsl: six 90-degree captures of a unit cube from its centre
 * (captureViews :160-167), fragment = texture(equirect, SampleSphericalMap(normalize(localPos))), equirect RGB16F
 * LINEAR / CLAMP_TO_EDGE (:126-132), faces RGB16F (:139-142; render-target stores round toward zero on the
 * reference's GL).  localPos is the interpolated cube position; at a pixel centre it is f + xn*s + yn*u with
 * (s, u, f) the right / up / forward vectors of that capture's lookAt (one component +-1, the others +-xn, +-yn).
 * atan / asin are Mesa's NIR lowerings (nir_builtin_builder.c: nir_atan2 / nir_atan polynomial, build_asin with
 * p0 = 0.086566724, p1 = -0.03102955), pinned bit for bit on llvmpipe by tests/golden/cubemap.npz's probes
 * (asin 100 %, atan2 99.95 %, uv 100 % of 8192 vectors); accurate libm versions differ from them by up to 4e-4 rad.
 * ------------------------------------------------------------------------------------------- */
static float mesa_atan(float yx) {
    const float a = fabsf(yx);
    const float t = fminf(a, 1.0f) / fmaxf(a, 1.0f);
    const float x2 = t * t, x3 = x2 * t, x5 = x3 * x2, x7 = x5 * x2, x9 = x7 * x2, x11 = x9 * x2;
    float p = t * 0.9999793128310355f;
    p = x3 * -0.3326756418091246f + p;
    p = x5 * 0.1938924977115610f + p;
    p = x7 * -0.1173503194786851f + p;
    p = x9 * 0.0536813784310406f + p;
    p = x11 * -0.0121323213173444f + p;
    p = p + (a > 1.0f ? 1.0f : 0.0f) * (p * -2.0f + 1.57079632679489661923f);
    return p * (yx > 0.0f ? 1.0f : yx < 0.0f ? -1.0f : 0.0f);
}
static float mesa_atan2(float y, float x) {
    const int flip = 0.0f >= x;
    const float s = flip ? fabsf(x) : y, t = flip ? y : fabsf(x);
    const float scale = fabsf(t) >= 1e18f ? 0.25f : 1.0f;
    const float rcp = 1.0f / (t * scale);
    const float s_over_t = (s * scale) * rcp;
    const float tn = fabsf(fabsf(x) == fabsf(y) ? 1.0f : s_over_t);
    const float arc = (flip ? 1.0f : 0.0f) * 1.57079632679489661923f + mesa_atan(tn);
    return fminf(y, rcp) < 0.0f ? -arc : arc;
}
static float mesa_asin(float x) {
    const float ax = fabsf(x);
    const float pi4m1 = 0.78539816339744830962f - 1.0f;      /* M_PI_4f - 1.0f, in float */
    float t = ax * -0.03102955f + 0.086566724f;
    t = ax * t + pi4m1;
    t = ax * t + 1.57079632679489661923f;
    const float r = 1.57079632679489661923f - sqrtf(1.0f - ax) * t;
    return (x > 0.0f ? 1.0f : x < 0.0f ? -1.0f : 0.0f) * r;
}
void orc_mesa_atan2_asin(const float *y, const float *x, const float *z, int n, float *outAtan2, float *outAsin) {
    for (int i = 0; i < n; i++) { outAtan2[i] = mesa_atan2(y[i], x[i]); outAsin[i] = mesa_asin(z[i]); }
}

/* equirect: W*H*3 floats as stbi_loadf returns them (row 0 = bottom after its flip); the RGB16F texture the
 * reference uploads them into stores them rounded toward zero on its GL (pinned by tests/golden/cubemap.npz's
 * upload probe); faces: 6*S*S*3 halfs, GL face order +X,-X,+Y,-Y,+Z,-Z, row 0 = t = 0. */
void orc_equirect_to_cubemap(const float *equirect, int W, int H, int S, uint16_t *faces) {
    static const float FS[6][3] = {{0, 0, -1}, {0, 0, 1}, {1, 0, 0}, {1, 0, 0}, {1, 0, 0}, {-1, 0, 0}};
    static const float FU[6][3] = {{0, -1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}, {0, -1, 0}, {0, -1, 0}};
    static const float FF[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
#pragma omp parallel for schedule(static) collapse(2)
    for (int f = 0; f < 6; f++)
        for (int j = 0; j < S; j++)
            for (int i = 0; i < S; i++) {
                const float xn = (((float)i + 0.5f) / (float)S) * 2.0f - 1.0f, yn = (((float)j + 0.5f) / (float)S) * 2.0f - 1.0f;
                float p[3], d[3];
                for (int c = 0; c < 3; c++) p[c] = FF[f][c] + xn * FS[f][c] + yn * FU[f][c];
                nrm3(p, d);                                                     /* skyboxFs.glsl:16 */
                float u = mesa_atan2(d[2], d[0]), v = mesa_asin(d[1]);          /* :9 */
                u = u * 0.1591f + 0.5f;                                         /* :10-11 */
                v = v * 0.3183f + 0.5f;
                /* LINEAR, CLAMP_TO_EDGE */
                const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
                const float fx = floorf(x), fy = floorf(y);
                const float wx = x - fx, wy = y - fy;
                int x0 = (int)fx, x1 = x0 + 1, y0 = (int)fy, y1 = y0 + 1;
                x0 = x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1);
                y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
                uint16_t *o = faces + (((size_t)f * S + j) * S + i) * 3;
                for (int c = 0; c < 3; c++) {
                    const float t00 = h2f(f2h_rt(equirect[((size_t)y0 * W + x0) * 3 + c])), t10 = h2f(f2h_rt(equirect[((size_t)y0 * W + x1) * 3 + c]));
                    const float t01 = h2f(f2h_rt(equirect[((size_t)y1 * W + x0) * 3 + c])), t11 = h2f(f2h_rt(equirect[((size_t)y1 * W + x1) * 3 + c]));
                    const float a = t00 + wx * (t10 - t00), b = t01 + wx * (t11 - t01);
                    o[c] = f2h_rt(a + wy * (b - a));
                }
            }
}
