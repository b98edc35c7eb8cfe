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
