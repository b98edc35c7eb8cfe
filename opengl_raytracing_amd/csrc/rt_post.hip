// rt_post.hip -- post passes that consume the ray tracer's surfaces, as gfx950 HIP kernels.
// First (and so far only) one: the TAA resolve of /root/reference/shader/taaFs.glsl:13-53, driven
// like /root/reference/src/ForwardShadingPipeline.cpp:231-260.  Unlike the ray tracer this IS a
// bandwidth-bound kernel: 56 B of compulsory HBM traffic per pixel (current 16 + history 16 +
// gNormal 8 read, 16 written) against ~150 flops.
//
// Layout: 256-thread workgroup = 64x4 pixels, one lane per pixel, 16-B loads and one float4 store per
// lane.  The 3x3 neighbourhood, the jittered bilinear tap, history's 4 taps (normally collapsing onto
// one texel) and the gNormal tap(s) are served by L1/L2 (RT_TAA_LDS=1 stages the current tile + halo in
// LDS instead: measured equal).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "rt_fastmath.h"

#ifndef RT_TAA_LDS
#define RT_TAA_LDS 0   // 1: stage the current-frame tile (+halo) in LDS; 0: neighbourhood straight from L1/L2.
#endif                // Measured equal within noise (29-31 us @1080p for every tile shape): the pass is limited by the
                      // memory pipeline, not by how the 3x3 taps are fetched; the simpler form is the default.  An XCD-aware
                      // tile order (one contiguous band of tiles per XCD, as in the bloom kernels) measured 3 % slower here.

namespace {

#ifndef RT_TAA_TX
#define RT_TAA_TX 64
#endif
constexpr int TX = RT_TAA_TX, TY = 256 / RT_TAA_TX, HALO = 1;
[[maybe_unused]] constexpr int LW = TX + 2 * HALO, LH = TY + 2 * HALO;

__device__ __forceinline__ int wrapi(int i, int n) { int r = i % n; return r < 0 ? r + n : r; }
__device__ __forceinline__ int clampi(int i, int n) { return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); }

struct rgb { float x, y, z; };
__device__ __forceinline__ rgb lerp3(rgb a, rgb b, float w) {   // a + w*(b-a), per llvmpipe's float path
    rgb r; r.x = a.x + w * (b.x - a.x); r.y = a.y + w * (b.y - a.y); r.z = a.z + w * (b.z - a.z); return r;
}

}  // namespace

__global__ __launch_bounds__(256) void rt_taa_resolve_kernel(const float4 *__restrict__ current,
                                                             const float4 *__restrict__ history,
                                                             const uint2 *__restrict__ normal,   // half4 per pixel
                                                             float4 *__restrict__ out, int W, int H, float blendFactor,
                                                             float jitterX, float jitterY) {
#if RT_TAA_LDS
    __shared__ float4 tile[LH][LW];
#endif
    const int bx = blockIdx.x * TX, by = blockIdx.y * TY;
    // stage the current-frame tile; texels outside the image are stored with REPEAT addressing (what the
    // bilinear tap needs); the 3x3 clamp below substitutes 0 for them itself (texelFetch out of range)
#if RT_TAA_LDS
    for (int k = threadIdx.x; k < LW * LH; k += 256) {
        const int lx = k % LW, ly = k / LW;
        const int gx = wrapi(bx + lx - HALO, W), gy = wrapi(by + ly - HALO, H);
        tile[ly][lx] = current[(size_t)gy * W + gx];
    }
    __syncthreads();
#endif
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int i = bx + tx, j = by + ty;
    if (i >= W || j >= H) return;

    auto cur_at = [&](int x, int y) -> rgb {   // REPEAT-addressed current texel (x, y may be any integer)
        const int lx = x - bx + HALO, ly = y - by + HALO;
        float4 q;
#if RT_TAA_LDS
        if (lx >= 0 && lx < LW && ly >= 0 && ly < LH) q = tile[ly][lx];
        else
#endif
        q = current[(size_t)wrapi(y, H) * W + wrapi(x, W)];
        (void)lx; (void)ly;
        rgb r; r.x = q.x; r.y = q.y; r.z = q.z; return r;
    };

    const float u = ((float)i + 0.5f) / (float)W, v = ((float)j + 0.5f) / (float)H;   // TexCoords
    const float ju = u + jitterX, jv = v + jitterY;                                     // taaFs.glsl:23
    // current = texture(uCurrentFrame, jitteredUV): LINEAR, REPEAT                     :24
    rgb cur;
    {
        const float x = ju * (float)W - 0.5f, y = jv * (float)H - 0.5f;
        const float fx = floorf(x), fy = floorf(y);
        const float wx = x - fx, wy = y - fy;
        const int x0 = (int)fx, y0 = (int)fy;
        rgb a = lerp3(cur_at(x0, y0), cur_at(x0 + 1, y0), wx);
        rgb b = lerp3(cur_at(x0, y0 + 1), cur_at(x0 + 1, y0 + 1), wx);
        cur = lerp3(a, b, wy);
    }
    // history = texture(uHistory, TexCoords): LINEAR, CLAMP_TO_EDGE                   :27
    rgb his;
    {
        const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
        const float fx = floorf(x), fy = floorf(y);
        const float wx = x - fx, wy = y - fy;
        const int x0 = clampi((int)fx, W), x1 = clampi((int)fx + 1, W), y0 = clampi((int)fy, H), y1 = clampi((int)fy + 1, H);
        auto h_at = [&](int xx, int yy) -> rgb { float4 q = history[(size_t)yy * W + xx]; rgb r; r.x = q.x; r.y = q.y; r.z = q.z; return r; };
        rgb a = lerp3(h_at(x0, y0), h_at(x1, y0), wx);
        rgb b = lerp3(h_at(x0, y1), h_at(x1, y1), wx);
        his = lerp3(a, b, wy);
    }
    // neighbourhood colour box, texelFetch (0 outside the image)                       :30-37
    rgb mn = cur, mx = cur;
#pragma unroll
    for (int dx = -1; dx <= 1; dx++) {
#pragma unroll
        for (int dy = -1; dy <= 1; dy++) {
            const int x = i + dx, y = j + dy;
            const bool inb = x >= 0 && y >= 0 && x < W && y < H;
#if RT_TAA_LDS
            const float4 q = tile[ty + dy + HALO][tx + dx + HALO];
#else
            const float4 q = current[(size_t)clampi(y, H) * W + clampi(x, W)];
#endif
            const float nx = inb ? q.x : 0.0f, ny = inb ? q.y : 0.0f, nz = inb ? q.z : 0.0f;
            mn.x = fminf(mn.x, nx); mn.y = fminf(mn.y, ny); mn.z = fminf(mn.z, nz);
            mx.x = fmaxf(mx.x, nx); mx.y = fmaxf(mx.y, ny); mx.z = fmaxf(mx.z, nz);
        }
    }
    // normal check, NEAREST / REPEAT                                                   :40-45
    float bf = 0.0f;
    {
        const int px = wrapi((int)floorf(u * (float)W), W), py = wrapi((int)floorf(v * (float)H), H);
        const int cx = wrapi((int)floorf(ju * (float)W), W), cy = wrapi((int)floorf(jv * (float)H), H);
        const uint2 pn = normal[(size_t)py * W + px];
        const uint2 cn = (cx == px && cy == py) ? pn : normal[(size_t)cy * W + cx];   // sub-texel jitter: same texel
        const float pnx = __half2float(__ushort_as_half((unsigned short)(pn.x & 0xffffu))), pny = __half2float(__ushort_as_half((unsigned short)(pn.x >> 16)));
        const float pnz = __half2float(__ushort_as_half((unsigned short)(pn.y & 0xffffu)));
        const float cnx = __half2float(__ushort_as_half((unsigned short)(cn.x & 0xffffu))), cny = __half2float(__ushort_as_half((unsigned short)(cn.x >> 16)));
        const float cnz = __half2float(__ushort_as_half((unsigned short)(cn.y & 0xffffu)));
        const float d = (pnz * cnz + pny * cny) + pnx * cnx;
        if (d < 0.9f) bf = blendFactor * 0.2f;
    }
    // clipAABB (:13-19) then mix(history, current, blendFactor) (:51)
    auto resolve = [&](float h, float c, float lo, float hi) -> float {
        const float center = 0.5f * (hi + lo), extents = 0.5f * (hi - lo);
        float clip = h - center;
        clip = fminf(fmaxf(clip, -extents), extents);
        const float hc = center + clip;
        return hc + bf * (c - hc);
    };
    out[(size_t)j * W + i] = make_float4(resolve(his.x, cur.x, mn.x, mx.x), resolve(his.y, cur.y, mn.y, mx.y),
                                         resolve(his.z, cur.z, mn.z, mx.z), 1.0f);
}

// The same resolve with a REGISTER WINDOW: one lane per column walks RT_TAA_ROWS consecutive rows and keeps three rows x three
// columns of `current` and of `history` in registers, so a pixel costs 6 x (ROWS + 2) / ROWS texel loads instead of 17.  The
// one-texel-per-lane kernel above is bound by the L1 pipeline rather than by HBM: its 17 loads per pixel are 288 B of L1 traffic
// against 56 B of compulsory bytes -- with the neighbourhood and bilinear loads stubbed out (timing experiment) it runs 20.3 us at
// 1080p against 28.7.  Arithmetic per pixel is the kernel above's, operation for operation; what changes is where a tap's texel
// comes from: the bilinear taps of `current` (REPEAT) are taken from the window when they fall inside it and inside the image (the
// jitter is a fraction of a texel: taaFs.glsl:23 with ForwardShadingPipeline.cpp:241-242) and loaded as before otherwise; the
// history taps (CLAMP_TO_EDGE) are window texels whenever floor(u * W - 0.5) is i - 1 or i, which fp32 guarantees for every pixel
// centre; the 3x3 box is the window itself (texelFetch: 0 outside the image).
// Measured, bit-identical (tools/bench_taa.py, same run): 1080p 28.8 -> 27.1 us, 4K 135 -> 118 us, 8K 498 -> 427 us.  Two
// further forms were written, pass the same tests and were NOT kept: each lane loading only its own column and taking the
// neighbouring columns from the adjacent lanes (DPP wave shifts), one row ahead of its use -- a quarter of the loads again, but
// 136 VGPRs / 3 waves and two loads in flight per wave: 36 / 139 / 531 us; and both surfaces staged as 64 x 8..32 pixel tiles in
// LDS (1.2 loads per pixel and surface, taps from LDS): 30 / 118 / 415 us at best.
#ifndef RT_TAA_ROWS
#define RT_TAA_ROWS 8
#endif
__global__ __launch_bounds__(256) void rt_taa_resolve_rows_kernel(const float4 *__restrict__ current, const float4 *__restrict__ history,
                                                                  const uint2 *__restrict__ normal, float4 *__restrict__ out, int W, int H,
                                                                  float blendFactor, float jitterX, float jitterY) {
    constexpr int R = RT_TAA_ROWS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const int j0 = (blockIdx.y * 4 + wave) * R;
    if (i >= W || j0 >= H) return;
    const int xl = clampi(i - 1, W), xr = clampi(i + 1, W);
    rgb c[3][3], h[3][3];            // [row slot][column: left, middle, right], rows / columns at CLAMPED addresses
    auto load_row = [&](int slot, int y) {
        const size_t o = (size_t)clampi(y, H) * W;
        const float4 c0 = current[o + xl], c1 = current[o + i], c2 = current[o + xr];
        const float4 h0 = history[o + xl], h1 = history[o + i], h2 = history[o + xr];
        c[slot][0] = {c0.x, c0.y, c0.z}; c[slot][1] = {c1.x, c1.y, c1.z}; c[slot][2] = {c2.x, c2.y, c2.z};
        h[slot][0] = {h0.x, h0.y, h0.z}; h[slot][1] = {h1.x, h1.y, h1.z}; h[slot][2] = {h2.x, h2.y, h2.z};
    };
    auto sel = [](bool first, rgb a, rgb b) -> rgb { rgb r; r.x = first ? a.x : b.x; r.y = first ? a.y : b.y; r.z = first ? a.z : b.z; return r; };
    load_row(0, j0 - 1);
    load_row(1, j0);
    const float u = ((float)i + 0.5f) / (float)W;
    const float ju = u + jitterX;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int j = j0 + r;
        if (j >= H) break;                                   // (wave-uniform)
        const int sa = r % 3, sm = (r + 1) % 3, sb = (r + 2) % 3;          // slots of rows j - 1, j, j + 1
        load_row(sb, j + 1);
        const float v = ((float)j + 0.5f) / (float)H;
        const float jv = v + jitterY;
        // current = texture(uCurrentFrame, jitteredUV): LINEAR, REPEAT                     :24
        rgb cur;
        {
            const float x = ju * (float)W - 0.5f, y = jv * (float)H - 0.5f;
            const float fx = floorf(x), fy = floorf(y);
            const float wx = x - fx, wy = y - fy;
            const int x0 = (int)fx, y0 = (int)fy;
            const int dx = x0 - i, dy = y0 - j;
            const bool inWin = (dx == -1 || dx == 0) && (dy == -1 || dy == 0) && x0 >= 0 && x0 + 1 < W && y0 >= 0 && y0 + 1 < H;
            const bool left = dx == -1, up = dy == -1;
            rgb t00 = sel(up, sel(left, c[sa][0], c[sa][1]), sel(left, c[sm][0], c[sm][1]));
            rgb t10 = sel(up, sel(left, c[sa][1], c[sa][2]), sel(left, c[sm][1], c[sm][2]));
            rgb t01 = sel(up, sel(left, c[sm][0], c[sm][1]), sel(left, c[sb][0], c[sb][1]));
            rgb t11 = sel(up, sel(left, c[sm][1], c[sm][2]), sel(left, c[sb][1], c[sb][2]));
            if (!inWin) {                                    // image border (REPEAT wraps around) or a jitter of a texel and more
                auto cur_at = [&](int xx, int yy) -> rgb { const float4 q = current[(size_t)wrapi(yy, H) * W + wrapi(xx, W)]; rgb t; t.x = q.x; t.y = q.y; t.z = q.z; return t; };
                t00 = cur_at(x0, y0); t10 = cur_at(x0 + 1, y0); t01 = cur_at(x0, y0 + 1); t11 = cur_at(x0 + 1, y0 + 1);
            }
            const rgb a = lerp3(t00, t10, wx), b = lerp3(t01, t11, wx);
            cur = lerp3(a, b, wy);
        }
        // history = texture(uHistory, TexCoords): LINEAR, CLAMP_TO_EDGE                   :27
        rgb his;
        {
            const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
            const float fx = floorf(x), fy = floorf(y);
            const float wx = x - fx, wy = y - fy;
            const int dx = (int)fx - i, dy = (int)fy - j;
            const bool inWin = (dx == -1 || dx == 0) && (dy == -1 || dy == 0);
            const bool left = dx == -1, up = dy == -1;
            // clamp(i + dx) and clamp(i + dx + 1) ARE the window's clamped columns (rows alike)
            rgb t00 = sel(up, sel(left, h[sa][0], h[sa][1]), sel(left, h[sm][0], h[sm][1]));
            rgb t10 = sel(up, sel(left, h[sa][1], h[sa][2]), sel(left, h[sm][1], h[sm][2]));
            rgb t01 = sel(up, sel(left, h[sm][0], h[sm][1]), sel(left, h[sb][0], h[sb][1]));
            rgb t11 = sel(up, sel(left, h[sm][1], h[sm][2]), sel(left, h[sb][1], h[sb][2]));
            if (!inWin) {
                const int x0 = clampi((int)fx, W), x1 = clampi((int)fx + 1, W), y0 = clampi((int)fy, H), y1 = clampi((int)fy + 1, H);
                auto h_at = [&](int xx, int yy) -> rgb { const float4 q = history[(size_t)yy * W + xx]; rgb t; t.x = q.x; t.y = q.y; t.z = q.z; return t; };
                t00 = h_at(x0, y0); t10 = h_at(x1, y0); t01 = h_at(x0, y1); t11 = h_at(x1, y1);
            }
            const rgb a = lerp3(t00, t10, wx), b = lerp3(t01, t11, wx);
            his = lerp3(a, b, wy);
        }
        // neighbourhood colour box, texelFetch (0 outside the image)                       :30-37
        rgb mn = cur, mx = cur;
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
#pragma unroll
            for (int dy = -1; dy <= 1; dy++) {
                const int x = i + dx, y = j + dy;
                const bool inb = x >= 0 && y >= 0 && x < W && y < H;
                const rgb q = c[dy < 0 ? sa : (dy == 0 ? sm : sb)][dx + 1];
                const float nx = inb ? q.x : 0.0f, ny = inb ? q.y : 0.0f, nz = inb ? q.z : 0.0f;
                mn.x = fminf(mn.x, nx); mn.y = fminf(mn.y, ny); mn.z = fminf(mn.z, nz);
                mx.x = fmaxf(mx.x, nx); mx.y = fmaxf(mx.y, ny); mx.z = fmaxf(mx.z, nz);
            }
        }
        // normal check, NEAREST / REPEAT                                                   :40-45
        float bf = 0.0f;
        {
            const int px = wrapi((int)floorf(u * (float)W), W), py = wrapi((int)floorf(v * (float)H), H);
            const int cx = wrapi((int)floorf(ju * (float)W), W), cy = wrapi((int)floorf(jv * (float)H), H);
            const uint2 pn = normal[(size_t)py * W + px];
            const uint2 cn = (cx == px && cy == py) ? pn : normal[(size_t)cy * W + cx];   // sub-texel jitter: same texel
            const float pnx = __half2float(__ushort_as_half((unsigned short)(pn.x & 0xffffu))), pny = __half2float(__ushort_as_half((unsigned short)(pn.x >> 16)));
            const float pnz = __half2float(__ushort_as_half((unsigned short)(pn.y & 0xffffu)));
            const float cnx = __half2float(__ushort_as_half((unsigned short)(cn.x & 0xffffu))), cny = __half2float(__ushort_as_half((unsigned short)(cn.x >> 16)));
            const float cnz = __half2float(__ushort_as_half((unsigned short)(cn.y & 0xffffu)));
            const float d = (pnz * cnz + pny * cny) + pnx * cnx;
            if (d < 0.9f) bf = blendFactor * 0.2f;
        }
        // clipAABB (:13-19) then mix(history, current, blendFactor) (:51)
        auto resolve = [&](float hh, float cc, float lo, float hi) -> float {
            const float center = 0.5f * (hi + lo), extents = 0.5f * (hi - lo);
            float clip = hh - center;
            clip = fminf(fmaxf(clip, -extents), extents);
            const float hc = center + clip;
            return hc + bf * (cc - hc);
        };
        out[(size_t)j * W + i] = make_float4(resolve(his.x, cur.x, mn.x, mx.x), resolve(his.y, cur.y, mn.y, mx.y),
                                             resolve(his.z, cur.z, mn.z, mx.z), 1.0f);
    }
}

#ifndef RT_TAA_WINDOW
#define RT_TAA_WINDOW 1         // 0: the one-texel-per-lane kernel for every launch
#endif
hipError_t rt_launch_taa_resolve(const void *current, const void *history, const void *normal, void *out, int W, int H,
                                 float blend, float jx, float jy, hipStream_t s) {
    // (`out` must not alias the inputs in either form; frames of a few rows keep the one-texel-per-lane kernel)
    if (RT_TAA_WINDOW && H >= 4 * RT_TAA_ROWS) {
        dim3 grid((W + 63) / 64, (H + 4 * RT_TAA_ROWS - 1) / (4 * RT_TAA_ROWS));
        hipLaunchKernelGGL(rt_taa_resolve_rows_kernel, grid, dim3(256), 0, s, (const float4 *)current, (const float4 *)history,
                           (const uint2 *)normal, (float4 *)out, W, H, blend, jx, jy);
        return hipGetLastError();
    }
    dim3 grid((W + TX - 1) / TX, (H + TY - 1) / TY);
    hipLaunchKernelGGL(rt_taa_resolve_kernel, grid, dim3(256), 0, s, (const float4 *)current, (const float4 *)history,
                       (const uint2 *)normal, (float4 *)out, W, H, blend, jx, jy);
    return hipGetLastError();
}

// =========================================================================================
// Bloom: brightness_extractFS.glsl, gaussian_blurFs.glsl, bloom_combineFs.glsl as
// driven by /root/reference/src/ForwardShadingPipeline.cpp:189-228.  Intermediate targets are rgba16f
// (:67-88) and the reference's GL rounds render-target writes toward zero, so every pass stores
// RTZ halfs; all taps sit on texel centres (LINEAR returns the texel), CLAMP_TO_EDGE at the borders.
// The chain runs as fused horizontal+vertical pairs (rt_bloom_hv_kernel, below); the single-pass kernels
// remain for odd / zero iteration counts.
// =========================================================================================
namespace {

// fp32 -> fp16 toward zero in hardware: v_cvt_pkrtz_f16_f32 is IEEE round-toward-zero (overflow saturates to
// +-65504, fp16 denormals kept -- the kernel runs with the default f16 denormal mode), i.e. the same function
// as the oracle's software orc_float_to_half_rtz on every non-NaN input; NaNs stay NaNs (payload not compared).
typedef __fp16 rt_h2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pkrtz_u(float lo, float hi) {
    union { rt_h2v h; unsigned u; } c;
    c.h = __builtin_amdgcn_cvt_pkrtz(lo, hi);
    return c.u;
}
__device__ __forceinline__ float h2f_u(unsigned h) { return __half2float(__ushort_as_half((unsigned short)(h & 0xffffu))); }
__device__ __forceinline__ uint2 pack_half4(float r, float g, float b) {
    return make_uint2(cvt_pkrtz_u(r, g), cvt_pkrtz_u(b, 1.0f));
}

}  // namespace

// brightness extract (brightness_extractFS.glsl:11-19)
__global__ __launch_bounds__(256) void rt_bloom_extract_kernel(const float4 *__restrict__ scene, uint2 *__restrict__ out,
                                                               size_t n, float threshold) {
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
        const float4 c = scene[k];
        const float brightness = (c.z * 0.0722f + c.y * 0.7152f) + c.x * 0.2126f;
        out[k] = (brightness > threshold) ? pack_half4(c.x, c.y, c.z) : make_uint2(0u, 0x3c00u << 16);
    }
}

// one separable 9-tap pass (gaussian_blurFs.glsl:8-26)
template <bool HORIZONTAL>
__global__ __launch_bounds__(256) void rt_bloom_blur_kernel(const uint2 *__restrict__ in, uint2 *__restrict__ out, int W, int H) {
    const float w0 = 0.227027f, w1 = 0.1945946f, w2 = 0.1216216f, w3 = 0.054054f, w4 = 0.016216f;
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= W || j >= H) return;
    auto tap = [&](int t) -> uint2 {
        const int x = HORIZONTAL ? min(max(i + t, 0), W - 1) : i, y = HORIZONTAL ? j : min(max(j + t, 0), H - 1);
        return in[(size_t)y * W + x];
    };
    const uint2 c = tap(0);
    float r = h2f_u(c.x) * w0, g = h2f_u(c.x >> 16) * w0, b = h2f_u(c.y) * w0;
    auto acc = [&](int t, float w) {
        const uint2 p = tap(t), m = tap(-t);
        r += h2f_u(p.x) * w; g += h2f_u(p.x >> 16) * w; b += h2f_u(p.y) * w;      // result += tex(+i)*w
        r += h2f_u(m.x) * w; g += h2f_u(m.x >> 16) * w; b += h2f_u(m.y) * w;      // result += tex(-i)*w
    };
    acc(1, w1); acc(2, w2); acc(3, w3); acc(4, w4);
    out[(size_t)j * W + i] = pack_half4(r, g, b);
}

// combine (bloom_combineFs.glsl:10-14)
__global__ __launch_bounds__(256) void rt_bloom_combine_kernel(const float4 *__restrict__ scene, const uint2 *__restrict__ bloom,
                                                               float4 *__restrict__ out, size_t n, float strength) {
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
        const float4 s = scene[k];
        const uint2 bl = bloom[k];
        out[k] = make_float4(s.x + h2f_u(bl.x) * strength, s.y + h2f_u(bl.x >> 16) * strength, s.z + h2f_u(bl.y) * strength, 1.0f);
    }
}

// Fused horizontal+vertical pass pair (one blur "iteration pair" of ForwardShadingPipeline.cpp:207-216):
// a 64x24-pixel workgroup tile stages its 72x32 input patch (4-texel apron each way, CLAMP_TO_EDGE applied
// while staging, in image space) in LDS as halfs, runs the horizontal pass into a second LDS array --
// rounded to fp16 toward zero exactly where the unfused chain stores its rgba16f target -- then the
// vertical pass from LDS.  Both passes use register sliding windows: a lane owns 4 consecutive outputs
// along the blur axis (3 for the vertical pass) on each of two lines, so it reads 3 (3.7) texels per output
// instead of 9, and the two lines share packed-fp32 arithmetic.  Lane->row mapping with odd-ish row strides (73 / 65 texels) keeps the
// ds_read_b64 / ds_write_b64 of 16 consecutive lanes on distinct banks.
// FIRST folds the brightness extract into the staging (reads the rgba32f scene), LAST folds the combine
// into the store (writes rgba32f).  HBM traffic per pair: 8 B in + 8 B out per pixel (+ apron re-reads
// served by L2) instead of 32 B for two unfused passes.
namespace {
constexpr int FX = 64, FY = 24, AP = 4, IW = FX + 2 * AP, IH = FY + 2 * AP;   // 72 x 32 patch
constexpr int IWP = IW + 1, FXP = FX + 1;                                     // padded LDS row strides
constexpr int HSPAN = 4, VSPAN = 3;                                           // outputs per lane and per half of the pair
static_assert((IH / 2) * (FX / HSPAN) == 256 && (FX / 2) * (FY / VSPAN) == 256, "one work item per thread in both passes");
typedef float rt_f2 __attribute__((ext_vector_type(2)));
}

// One 9-tap window on TWO independent lines at once (.x / .y): every multiply and add is a packed fp32
// instruction (v_pk_mul_f32 / v_pk_add_f32: IEEE, no contraction), which halves the VALU work this
// VALU-bound kernel issues; the order of operations per line is the reference's.
template <int N>
__device__ __forceinline__ void blur_window2(const rt_f2 (&r)[N + 8], const rt_f2 (&g)[N + 8], const rt_f2 (&b)[N + 8], int k,
                                             rt_f2 &orr, rt_f2 &og, rt_f2 &ob) {
    const float w[5] = {0.227027f, 0.1945946f, 0.1216216f, 0.054054f, 0.016216f};
    const int c = k + 4;
    rt_f2 ar = r[c] * w[0], ag = g[c] * w[0], ab = b[c] * w[0];  // gaussian_blurFs.glsl:13
#pragma unroll
    for (int t = 1; t < 5; t++) {                                 // :15-24: result += tex(+t)*w; result += tex(-t)*w
        ar += r[c + t] * w[t]; ag += g[c + t] * w[t]; ab += b[c + t] * w[t];
        ar += r[c - t] * w[t]; ag += g[c - t] * w[t]; ab += b[c - t] * w[t];
    }
    orr = ar; og = ag; ob = ab;
}

#ifndef RT_BLOOM_XCD
#define RT_BLOOM_XCD 1      // XCD-aware tile order of the fused kernels (needs RT_BLOOM_TPW == 1)
#endif
#ifndef RT_BLOOM_TPW
#define RT_BLOOM_TPW 1      // tiles per workgroup of the fused kernels (see rt_bloom_hv_kernel)
#endif
template <bool FIRST, bool LAST>
__global__ __launch_bounds__(256) void rt_bloom_hv_kernel(const void *__restrict__ inV, const float4 *__restrict__ scene,
                                                          void *__restrict__ outV, int W, int H, float threshold, float strength) {
    // one LDS array, used twice: the input patch (halfs; patch (r,c) <-> image (clamp(y0+r-4), clamp(x0+c-4))),
    // then -- once every lane holds its window in registers -- the horizontal-pass result (halfs, RTZ) for
    // patch rows 0..IH-1, tile columns 0..FX-1.  18.7 KB per workgroup: 8 workgroups (32 waves) per CU.
    __shared__ uint2 sin_[IH * IWP];
    uint2 *sh_ = sin_;
    // The patch is REQUESTED as one batch of NLD independent loads per thread into registers and written to LDS afterwards
    // (round 1 interleaved load -> ds_write per texel, which serialised the memory latency: 88 -> 72 us for the 1080p chain,
    // 249 -> 226 us at 4K).  The tile loop also lets a workgroup walk several tiles with the next patch in flight during
    // the two passes (RT_BLOOM_TPW tiles per workgroup): measured SLOWER -- 72 / 79 / 81 / 96 / 104 us at 1, 2, 3, 4, 6
    // tiles per workgroup at 1080p -- the 1 350 one-tile workgroups are all co-resident and cover each other's phases
    // better than fewer, longer-lived ones; the default stays one tile per workgroup.
    const int tilesX = (W + FX - 1) / FX, nTiles = tilesX * ((H + FY - 1) / FY);
    constexpr int NLD = (IW * IH) / 256;                               // patch texels per thread (9)
    static_assert(NLD * 256 == IW * IH, "whole patch in NLD loads per thread");
    float4 preF[FIRST ? NLD : 1];                                      // FIRST: the rgba32f scene texel (extract happens at the LDS write)
    uint2 preH[FIRST ? 1 : NLD];
    auto request = [&](int tile) {
        const int x0 = (tile % tilesX) * FX, y0 = (tile / tilesX) * FY;
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int k = threadIdx.x + j * 256;
            const int lx = k % IW, ly = k / IW;
            const int gx = min(max(x0 + lx - AP, 0), W - 1), gy = min(max(y0 + ly - AP, 0), H - 1);
            if (FIRST) preF[j] = scene[(size_t)gy * W + gx];
            else preH[j] = ((const uint2 *)inV)[(size_t)gy * W + gx];
        }
    };
#if RT_BLOOM_XCD
    // workgroups go to the 8 XCDs round-robin by id: give each XCD one contiguous band of tiles, so that the aprons a tile shares
    // with its neighbours are served by the SAME L2 instead of being fetched from HBM once per XCD
    int tile = (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
#else
    int tile = blockIdx.x;
#endif
    if (tile < nTiles) request(tile);
    for (; tile < nTiles; tile += (RT_BLOOM_TPW == 1 ? nTiles : (int)gridDim.x)) {      // (one trip when RT_BLOOM_TPW == 1)
    const int x0 = (tile % tilesX) * FX, y0 = (tile / tilesX) * FY;
    // ---- the requested patch -> LDS
#pragma unroll
    for (int j = 0; j < NLD; j++) {
        const int k = threadIdx.x + j * 256;
        uint2 v;
        if (FIRST) {     // brightness_extractFS.glsl:11-19 on the fly
            const float4 c = preF[j];
            const float brightness = (c.z * 0.0722f + c.y * 0.7152f) + c.x * 0.2126f;
            v = (brightness > threshold) ? pack_half4(c.x, c.y, c.z) : make_uint2(0u, 0x3c00u << 16);
        } else {
            v = preH[j];
        }
        sin_[(k / IW) * IWP + (k % IW)] = v;
    }
    __syncthreads();
    if (RT_BLOOM_TPW > 1 && tile + (int)gridDim.x < nTiles) request(tile + (int)gridDim.x);     // in flight during the two passes below
    {   // ---- horizontal pass: lane -> patch rows (rp, rp+16), 4 consecutive columns of each
        const int rp = threadIdx.x % (IH / 2), grp = threadIdx.x / (IH / 2);
        rt_f2 r[HSPAN + 8], g[HSPAN + 8], b[HSPAN + 8];
#pragma unroll
        for (int c = 0; c < HSPAN + 8; c++) {
            const uint2 v0 = sin_[rp * IWP + grp * HSPAN + c], v1 = sin_[(rp + IH / 2) * IWP + grp * HSPAN + c];
            r[c].x = h2f_u(v0.x); g[c].x = h2f_u(v0.x >> 16); b[c].x = h2f_u(v0.y);
            r[c].y = h2f_u(v1.x); g[c].y = h2f_u(v1.x >> 16); b[c].y = h2f_u(v1.y);
        }
        __syncthreads();     // every window is in registers: the patch storage may be overwritten
#pragma unroll
        for (int k = 0; k < HSPAN; k++) {
            rt_f2 orr, og, ob;
            blur_window2<HSPAN>(r, g, b, k, orr, og, ob);
            sh_[rp * FXP + grp * HSPAN + k] = pack_half4(orr.x, og.x, ob.x);
            sh_[(rp + IH / 2) * FXP + grp * HSPAN + k] = pack_half4(orr.y, og.y, ob.y);
        }
    }
    __syncthreads();
    {   // ---- vertical pass: lane -> columns (lx, lx+32), 3 consecutive rows of each
        const int lx = threadIdx.x % (FX / 2), seg = threadIdx.x / (FX / 2);
        rt_f2 r[VSPAN + 8], g[VSPAN + 8], b[VSPAN + 8];
#pragma unroll
        for (int c = 0; c < VSPAN + 8; c++) {
            const uint2 v0 = sh_[(seg * VSPAN + c) * FXP + lx], v1 = sh_[(seg * VSPAN + c) * FXP + lx + FX / 2];
            r[c].x = h2f_u(v0.x); g[c].x = h2f_u(v0.x >> 16); b[c].x = h2f_u(v0.y);
            r[c].y = h2f_u(v1.x); g[c].y = h2f_u(v1.x >> 16); b[c].y = h2f_u(v1.y);
        }
        __syncthreads();     // the horizontal result is in registers: the next tile's patch may overwrite the LDS array
#pragma unroll
        for (int k = 0; k < VSPAN; k++) {
            const int y = y0 + seg * VSPAN + k;
            if (y >= H) continue;
            rt_f2 orr, og, ob;
            blur_window2<VSPAN>(r, g, b, k, orr, og, ob);
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int x = x0 + lx + half * (FX / 2);
                if (x >= W) continue;
                const uint2 hv = half ? pack_half4(orr.y, og.y, ob.y) : pack_half4(orr.x, og.x, ob.x);   // the rgba16f store
                if (LAST) {      // bloom_combineFs.glsl:10-14
                    const float4 sc = scene[(size_t)y * W + x];
                    ((float4 *)outV)[(size_t)y * W + x] = make_float4(sc.x + h2f_u(hv.x) * strength, sc.y + h2f_u(hv.x >> 16) * strength,
                                                                      sc.z + h2f_u(hv.y) * strength, 1.0f);
                } else {
                    ((uint2 *)outV)[(size_t)y * W + x] = hv;
                }
            }
        }
    }
    }   // tiles of this workgroup
}

hipError_t rt_launch_bloom(const void *scene, void *tmpA, void *tmpB, void *out, int W, int H, float threshold, float strength,
                           int iterations, hipStream_t s) {
    const size_t n = (size_t)W * H;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    uint2 *a = (uint2 *)tmpA, *b = (uint2 *)tmpB;
    const float4 *sc = (const float4 *)scene;
    const int pairs = iterations / 2;
    const bool odd = (iterations & 1) != 0;
    const int nTilesF = ((W + FX - 1) / FX) * ((H + FY - 1) / FY);
    dim3 fgrid(RT_BLOOM_XCD ? ((nTilesF + 7) / 8) * 8 : (nTilesF + RT_BLOOM_TPW - 1) / RT_BLOOM_TPW), grid((W + 63) / 64, (H + 3) / 4);
    if (pairs == 0) {            // 0 or 1 iterations: the unfused kernels
        hipLaunchKernelGGL(rt_bloom_extract_kernel, dim3(blocks), dim3(256), 0, s, sc, a, n, threshold);
        if (odd) { hipLaunchKernelGGL(rt_bloom_blur_kernel<true>, grid, dim3(256), 0, s, a, b, W, H); a = b; }
        hipLaunchKernelGGL(rt_bloom_combine_kernel, dim3(blocks), dim3(256), 0, s, sc, a, (float4 *)out, n, strength);
        return hipGetLastError();
    }
    for (int p = 0; p < pairs; p++) {
        const bool first = p == 0, last = (p == pairs - 1) && !odd;
        void *dst = last ? out : (void *)b;
        if (first && last) hipLaunchKernelGGL((rt_bloom_hv_kernel<true, true>), fgrid, dim3(256), 0, s, (const void *)a, sc, dst, W, H, threshold, strength);
        else if (first) hipLaunchKernelGGL((rt_bloom_hv_kernel<true, false>), fgrid, dim3(256), 0, s, (const void *)a, sc, dst, W, H, threshold, strength);
        else if (last) hipLaunchKernelGGL((rt_bloom_hv_kernel<false, true>), fgrid, dim3(256), 0, s, (const void *)a, sc, dst, W, H, threshold, strength);
        else hipLaunchKernelGGL((rt_bloom_hv_kernel<false, false>), fgrid, dim3(256), 0, s, (const void *)a, sc, dst, W, H, threshold, strength);
        if (!last) { uint2 *t = a; a = b; b = t; }
    }
    if (odd) {                   // trailing horizontal pass, then the combine
        hipLaunchKernelGGL(rt_bloom_blur_kernel<true>, grid, dim3(256), 0, s, a, b, W, H);
        hipLaunchKernelGGL(rt_bloom_combine_kernel, dim3(blocks), dim3(256), 0, s, sc, b, (float4 *)out, n, strength);
    }
    return hipGetLastError();
}

// =========================================================================================
// SSAO: ssaoFs.glsl:16-46 + ssao_blurFs.glsl:11-29 as driven by
// /root/reference/src/AO.cpp:86-117.  G-buffer consumer: gPosition (rgba32f) and gNormal (rgba16f) as the ray
// kernel wrote them, NEAREST with the default REPEAT wrap (ForwardShadingPipeline.cpp:115-126).  Same fp32
// expression shapes as oracle/rt_post_oracle.c::orc_ssao (pinned bit for bit to the shader on llvmpipe).
// MI355X mapping: the 64 kernel samples, both matrices and the 4x4 rotation texture are kernel arguments
// (1.2 KB): wave-uniform, so they arrive as SGPR operands by s_load; a wave owns an 8x8 pixel tile so its
// 64 x 64 depth gathers (4 B out of every 16 B texel) stay in a few L1/L2 lines.
// =========================================================================================
#ifndef RT_SSAO_UNROLL
#define RT_SSAO_UNROLL 8
#endif
#ifndef RT_SSAO_XCD
#define RT_SSAO_XCD 0             // 1: one contiguous image band per XCD.  Measured neutral at 1080p (387 vs 386 us) and
#endif                            // 1.5 % slower at 4K: the kernel is VALU-bound, not L2-bound; kept as a switch.
#ifndef RT_SSAO_TILE_W
#define RT_SSAO_TILE_W 8          // wave tile width in pixels (height = 64 / width)
#endif
struct RtSsaoArgs {
    float samples[64][3];
    float projection[16], view[16];
    float noise[16][4];
    int nW, nH, W, H;
};

namespace {
__device__ __forceinline__ int ssao_nearest_repeat(float u, int size) {
    if ((size & (size - 1)) == 0) return ((int)floorf(u * (float)size)) & (size - 1);
    float fr = u - floorf(u);
    if (!(fr < 1.0f)) fr = 0.99999994f;
    if (!(fr >= 0.0f)) fr = 0.0f;
    return (int)(fr * (float)size);
}
__device__ __forceinline__ void ssao_nrm3(float x, float y, float z, float &ox, float &oy, float &oz) {
    const float d = (z * z + y * y) + x * x;
    const float r = 1.0f / sqrtf(d);
    ox = x * r; oy = y * r; oz = z * r;
}
}  // namespace

// gPosition.z as a dense plane: the 64 depth gathers per pixel then touch 4x fewer cache lines
__global__ __launch_bounds__(256) void rt_ssao_depth_kernel(const float4 *__restrict__ position, float *__restrict__ depth, size_t n) {
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) depth[k] = position[k].z;
}

#ifndef RT_SSAO_FASTDIV
#define RT_SSAO_FASTDIV 1       // 0 = the compiler's IEEE divisions in the sample loop
#endif

// ssaoFs.glsl:27-44, the 64 kernel samples of one pixel.  FAST: divisions by rt_fastmath.h's fast paths; returns NaN when
// any of them left its exact range (the caller then takes the IEEE instantiation).  The true sum is never NaN-free-checked:
// a genuine NaN (NaN G-buffer) also sends the wave through the IEEE loop, which reproduces it.
template <bool FAST>
__device__ __forceinline__ float ssao_samples(const RtSsaoArgs &a, const float *__restrict__ depth, const float4 fp, float nx, float ny, float nz,
                                              float tx, float ty, float tz, float bx, float by, float bz, int W, int H) {
    float occlusion = 0.0f;
    bool bad = false;
    const float *V = a.view, *P = a.projection;
#pragma unroll RT_SSAO_UNROLL
    for (int k = 0; k < 64; k++) {
        const float s0 = a.samples[k][0], s1 = a.samples[k][1], s2 = a.samples[k][2];
        float px = (tx * s0 + bx * s1) + nx * s2, py = (ty * s0 + by * s1) + ny * s2, pz = (tz * s0 + bz * s1) + nz * s2;
        px = fp.x + px * 0.5f; py = fp.y + py * 0.5f; pz = fp.z + pz * 0.5f;
        float vw[4], of[4];
#pragma unroll
        for (int r = 0; r < 4; r++) vw[r] = ((V[r] * px + V[4 + r] * py) + V[8 + r] * pz) + V[12 + r] * 1.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) of[r] = ((P[r] * vw[0] + P[4 + r] * vw[1]) + P[8 + r] * vw[2]) + P[12 + r] * vw[3];
        float ox, oy;
        if constexpr (FAST) {           // offset.xy / offset.w: one refined reciprocal, two Markstein corrections
            bool oky, ok0, ok1;
            const float y = rtf::rcp_fast(of[3], oky);
            ox = rtf::div_fast(of[0], of[3], y, ok0);
            oy = rtf::div_fast(of[1], of[3], y, ok1);
            bad |= !(oky & ok0 & ok1);
        } else {
            ox = of[0] / of[3]; oy = of[1] / of[3];
        }
        ox = ox * 0.5f + 0.5f;
        oy = oy * 0.5f + 0.5f;
        const float sampleDepth = depth[(size_t)ssao_nearest_repeat(oy, H) * W + ssao_nearest_repeat(ox, W)];
        const float adz = fabsf(fp.z - sampleDepth);
        float x;
        if constexpr (FAST) {           // 0.5 / |dz| = RN(1 / |dz|) * 0.5: the halving is exact while reciprocal and product are normal
            bool okr;
            x = rtf::rcp_fast(adz, okr) * 0.5f;
            const bool zero = adz == 0.0f;        // equal depths (flat regions, the pixel itself): 0.5 / 0 = +inf, no reason to leave the fast loop
            x = zero ? __builtin_huge_valf() : x;
            bad |= !((okr & (adz <= 0x1p125f)) | zero);
        } else {
            x = 0.5f / adz;
        }
        const float tt = fminf(fmaxf(x, 0.0f), 1.0f);
        const float rangeCheck = tt * (tt * (3.0f - 2.0f * tt));
        occlusion += (sampleDepth >= pz + 0.025f ? 1.0f : 0.0f) * rangeCheck;
    }
    return (FAST && bad) ? __builtin_nanf("") : occlusion;
}

__global__ __launch_bounds__(256) void rt_ssao_kernel(const float4 *__restrict__ position, const uint2 *__restrict__ normal,
                                                      const float *__restrict__ depth, float *__restrict__ out, const RtSsaoArgs a) {
    // 256 threads = 4 waves, each an 8x8 tile of a 32x8 block
    constexpr int TW = RT_SSAO_TILE_W, TH = 64 / TW;
    const int wave = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int W = a.W, H = a.H;
    // XCD-aware block -> tile map: workgroup b runs on XCD b % 8, so give each XCD one contiguous band of the image
    // (its 4 MB L2 then holds the band's depth neighbourhood instead of 1/8 of every neighbourhood of the frame)
    const int nbx = (W + 4 * TW - 1) / (4 * TW);
#if RT_SSAO_XCD
    const unsigned nb = gridDim.x, per = (nb + 7u) / 8u;
    const unsigned lb = (blockIdx.x % 8u) * per + blockIdx.x / 8u;
    if (lb >= nb) return;            // nb not a multiple of 8: the map below covers [0, nb) exactly once (see launch)
#else
    const unsigned lb = blockIdx.x;
#endif
    const int blkX = (int)(lb % (unsigned)nbx), blkY = (int)(lb / (unsigned)nbx);
    const int i = blkX * (4 * TW) + wave * TW + (ln % TW), j = blkY * TH + (ln / TW);
    if (i >= W || j >= H) return;
    const float u = ((float)i + 0.5f) / (float)W, v = ((float)j + 0.5f) / (float)H;
    const size_t self = (size_t)ssao_nearest_repeat(v, H) * W + ssao_nearest_repeat(u, W);
    const float4 fp = position[self];
    const uint2 nh = normal[self];
    float nx, ny, nz, rx, ry, rz;
    ssao_nrm3(h2f_u(nh.x), h2f_u(nh.x >> 16), h2f_u(nh.y), nx, ny, nz);
    const float nu = u * 200.0f, nv = v * 200.0f;                                       // ssaoFs.glsl:14,20
    const int nt = ssao_nearest_repeat(nv, a.nH) * a.nW + ssao_nearest_repeat(nu, a.nW);
    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; k++)       // per-lane pick from the SGPR-resident texture (<= 16 texels)
        if (k == nt) { qx = a.noise[k][0]; qy = a.noise[k][1]; qz = a.noise[k][2]; }
    ssao_nrm3(qx, qy, qz, rx, ry, rz);
    const float d = (rz * nz + ry * ny) + rx * nx;
    float tx, ty, tz;
    ssao_nrm3(rx - nx * d, ry - ny * d, rz - nz * d, tx, ty, tz);
    const float bx = ny * tz - ty * nz, by = nz * tx - tz * nx, bz = nx * ty - tx * ny;
    // The three divisions per sample (offset.xy / offset.w, 0.5 / |dz|) without the compiler's 11-instruction IEEE sequences
    // (rt_fastmath.h: correctly rounded for operands in the ordinary exponent range, each with an `ok` flag).  The sample loop
    // stays BRANCH-FREE -- a wave-uniform fallback branch per sample keeps the unrolled iterations' depth gathers from
    // overlapping (measured: 398 -> 449 us @1080p with it, i.e. slower than the IEEE divisions) -- so the flags are only
    // OR-ed, and a wave in which any lane's any sample left the fast range redoes its 64 samples with the IEEE divisions.
    // (a wave holding a pixel without geometry -- zero normal, NaN basis: the sky -- goes to the IEEE loop directly)
    const float chk = ((tx + ty) + tz) + ((bx + by) + bz) + ((fp.x + fp.y) + fp.z);
    float occlusion;
    if (RT_SSAO_FASTDIV != 0 && __builtin_amdgcn_ballot_w64(!(fabsf(chk) < __builtin_huge_valf())) == 0ull) {
        occlusion = ssao_samples<true>(a, depth, fp, nx, ny, nz, tx, ty, tz, bx, by, bz, W, H);
        if (__builtin_amdgcn_ballot_w64(occlusion != occlusion) != 0ull)
            occlusion = ssao_samples<false>(a, depth, fp, nx, ny, nz, tx, ty, tz, bx, by, bz, W, H);
    } else {
        occlusion = ssao_samples<false>(a, depth, fp, nx, ny, nz, tx, ty, tz, bx, by, bz, W, H);
    }
    out[(size_t)j * W + i] = 1.0f - occlusion / 64.0f;
}

template <bool HORIZONTAL>
__global__ __launch_bounds__(256) void rt_ssao_blur_kernel(const float *__restrict__ in, float *__restrict__ out, int W, int H) {
    const float wgt[5] = {0.227027f, 0.1945946f, 0.1216216f, 0.054054f, 0.016216f};
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= W || j >= H) return;
    const float u = ((float)i + 0.5f) / (float)W, v = ((float)j + 0.5f) / (float)H;
    const float tx = 1.0f / (float)W, ty = 1.0f / (float)H;
    float r = in[(size_t)ssao_nearest_repeat(v, H) * W + ssao_nearest_repeat(u, W)] * wgt[0];
#pragma unroll
    for (int k = 1; k < 5; k++) {
        const float du = HORIZONTAL ? tx * (float)k : 0.0f, dv = HORIZONTAL ? 0.0f : ty * (float)k;
        r += in[(size_t)ssao_nearest_repeat(v + dv, H) * W + ssao_nearest_repeat(u + du, W)] * wgt[k];
        r += in[(size_t)ssao_nearest_repeat(v - dv, H) * W + ssao_nearest_repeat(u - du, W)] * wgt[k];
    }
    out[(size_t)j * W + i] = r;
}

hipError_t rt_launch_ssao(const void *position, const void *normal, void *depthPlane, void *out, int W, int H, const float *noise,
                          int nW, int nH, const float *samples, const float *projection, const float *view, hipStream_t s) {
    RtSsaoArgs a;
    memset(&a, 0, sizeof a);
    memcpy(a.samples, samples, sizeof a.samples);
    memcpy(a.projection, projection, sizeof a.projection);
    memcpy(a.view, view, sizeof a.view);
    memcpy(a.noise, noise, (size_t)nW * nH * 4 * sizeof(float));
    a.nW = nW; a.nH = nH; a.W = W; a.H = H;
    constexpr int TW = RT_SSAO_TILE_W, TH = 64 / TW;
    unsigned nBlocks = (unsigned)(((W + 4 * TW - 1) / (4 * TW)) * ((H + TH - 1) / TH));
#if RT_SSAO_XCD
    nBlocks = (nBlocks + 7u) / 8u * 8u;      // whole groups of 8 so that b -> (b % 8) * per + b / 8 is a bijection
#endif
    dim3 grid(nBlocks);
    const size_t n = (size_t)W * H;
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(rt_ssao_depth_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float4 *)position, (float *)depthPlane, n);
    hipLaunchKernelGGL(rt_ssao_kernel, grid, dim3(256), 0, s, (const float4 *)position, (const uint2 *)normal,
                       (const float *)depthPlane, (float *)out, a);
    return hipGetLastError();
}

hipError_t rt_launch_ssao_blur(const void *in, void *out, int W, int H, int horizontal, hipStream_t s) {
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    if (horizontal) hipLaunchKernelGGL(rt_ssao_blur_kernel<true>, grid, dim3(256), 0, s, (const float *)in, (float *)out, W, H);
    else hipLaunchKernelGGL(rt_ssao_blur_kernel<false>, grid, dim3(256), 0, s, (const float *)in, (float *)out, W, H);
    return hipGetLastError();
}

// =========================================================================================
// Equirectangular -> cubemap: ConvertHDRToCubemap (TextureLoader.cpp:118-194) with
// skyboxVs.glsl / skyboxFs.glsl -- six 90-degree captures of a unit cube, per texel
// texture(equirect, SampleSphericalMap(normalize(localPos))).  Same fp32 expression shapes as
// oracle/rt_post_oracle.c::orc_equirect_to_cubemap, including Mesa's atan2 / asin lowerings (the oracle is
// pinned to the shader on llvmpipe with them).  One-off producer of the ray kernel's skybox (6 x 512^2 texels):
// one lane per face texel, the equirect map is converted to RGB16F once (what GL's upload does, toward zero
// on the reference's GL) and gathered with 4 bilinear taps; faces are written as packed RGB16F, round toward zero.
// =========================================================================================
namespace {
__device__ __forceinline__ float cm_sign(float x) { return x > 0.0f ? 1.0f : x < 0.0f ? -1.0f : 0.0f; }
__device__ __forceinline__ float cm_atan(float yx) {
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
    return p * cm_sign(yx);
}
__device__ __forceinline__ float cm_atan2(float y, float x) {
    const bool flip = 0.0f >= x;
    const float s = flip ? fabsf(x) : y, t = flip ? y : fabsf(x);
    const float scale = fabsf(t) >= 1e18f ? 0.25f : 1.0f;
    const float rcp = 1.0f / (t * scale);
    const float s_over_t = (s * scale) * rcp;
    const float tn = fabsf(fabsf(x) == fabsf(y) ? 1.0f : s_over_t);
    const float arc = (flip ? 1.0f : 0.0f) * 1.57079632679489661923f + cm_atan(tn);
    return fminf(y, rcp) < 0.0f ? -arc : arc;
}
__device__ __forceinline__ float cm_asin(float x) {
    const float ax = fabsf(x);
    const float pi4m1 = 0.78539816339744830962f - 1.0f;
    float t = ax * -0.03102955f + 0.086566724f;
    t = ax * t + pi4m1;
    t = ax * t + 1.57079632679489661923f;
    const float r = 1.57079632679489661923f - sqrtf(1.0f - ax) * t;
    return cm_sign(x) * r;
}
__constant__ float CM_S[6][3] = {{0, 0, -1}, {0, 0, 1}, {1, 0, 0}, {1, 0, 0}, {1, 0, 0}, {-1, 0, 0}};
__constant__ float CM_U[6][3] = {{0, -1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}, {0, -1, 0}, {0, -1, 0}};
__constant__ float CM_F[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
}  // namespace

// f32 RGB -> RGB16F texels (glTexImage2D(GL_RGB16F, GL_FLOAT) upload: the reference's GL rounds toward zero there
// too -- tests/golden/cubemap.npz's upload probe), padded to 4 halfs
__global__ __launch_bounds__(256) void rt_equirect_upload_kernel(const float *__restrict__ rgb, uint2 *__restrict__ tex, size_t n) {
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
        tex[k] = make_uint2(cvt_pkrtz_u(rgb[3 * k], rgb[3 * k + 1]), cvt_pkrtz_u(rgb[3 * k + 2], 1.0f));
    }
}

__global__ __launch_bounds__(256) void rt_equirect_to_cubemap_kernel(const uint2 *__restrict__ tex, int W, int H, int S,
                                                                     unsigned short *__restrict__ faces) {
    const int i = blockIdx.x * 32 + (threadIdx.x & 31), j = blockIdx.y * 8 + (threadIdx.x >> 5), f = blockIdx.z;
    if (i >= S || j >= S) return;
    const float xn = (((float)i + 0.5f) / (float)S) * 2.0f - 1.0f, yn = (((float)j + 0.5f) / (float)S) * 2.0f - 1.0f;
    const float px = CM_F[f][0] + xn * CM_S[f][0] + yn * CM_U[f][0], py = CM_F[f][1] + xn * CM_S[f][1] + yn * CM_U[f][1];
    const float pz = CM_F[f][2] + xn * CM_S[f][2] + yn * CM_U[f][2];
    const float dd = (pz * pz + py * py) + px * px;
    const float r = 1.0f / sqrtf(dd);
    const float dx = px * r, dy = py * r, dz = pz * r;                           // normalize(localPos)
    float u = cm_atan2(dz, dx), v = cm_asin(dy);                                 // SampleSphericalMap
    u = u * 0.1591f + 0.5f;
    v = v * 0.3183f + 0.5f;
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;                // LINEAR, CLAMP_TO_EDGE
    const float fx = floorf(x), fy = floorf(y);
    const float wx = x - fx, wy = y - fy;
    const int x0 = min(max((int)fx, 0), W - 1), x1 = min(max((int)fx + 1, 0), W - 1);
    const int y0 = min(max((int)fy, 0), H - 1), y1 = min(max((int)fy + 1, 0), H - 1);
    const uint2 t00 = tex[(size_t)y0 * W + x0], t10 = tex[(size_t)y0 * W + x1], t01 = tex[(size_t)y1 * W + x0], t11 = tex[(size_t)y1 * W + x1];
    auto lerp2 = [&](float a00, float a10, float a01, float a11) -> float {
        const float a = a00 + wx * (a10 - a00), b = a01 + wx * (a11 - a01);
        return a + wy * (b - a);
    };
    const float cr = lerp2(h2f_u(t00.x), h2f_u(t10.x), h2f_u(t01.x), h2f_u(t11.x));
    const float cg = lerp2(h2f_u(t00.x >> 16), h2f_u(t10.x >> 16), h2f_u(t01.x >> 16), h2f_u(t11.x >> 16));
    const float cb = lerp2(h2f_u(t00.y), h2f_u(t10.y), h2f_u(t01.y), h2f_u(t11.y));
    const unsigned rg = cvt_pkrtz_u(cr, cg), b1 = cvt_pkrtz_u(cb, 1.0f);
    unsigned short *o = faces + (((size_t)f * S + j) * S + i) * 3;
    o[0] = (unsigned short)(rg & 0xffffu); o[1] = (unsigned short)(rg >> 16); o[2] = (unsigned short)(b1 & 0xffffu);
}

hipError_t rt_launch_equirect_to_cubemap(const float *dRgb, void *dTex, int W, int H, int S, void *dFaces, hipStream_t s) {
    const size_t n = (size_t)W * H;
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(rt_equirect_upload_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dRgb, (uint2 *)dTex, n);
    dim3 grid((S + 31) / 32, (S + 7) / 8, 6);
    hipLaunchKernelGGL(rt_equirect_to_cubemap_kernel, grid, dim3(256), 0, s, (const uint2 *)dTex, W, H, S, (unsigned short *)dFaces);
    return hipGetLastError();
}
