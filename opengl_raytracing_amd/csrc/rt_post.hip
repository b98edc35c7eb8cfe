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

#ifndef RT_TAA_LDS
#define RT_TAA_LDS 0   // 1: stage the current-frame tile (+halo) in LDS; 0: neighbourhood straight from L1/L2.
#endif                // Measured equal within noise (29-31 us @1080p for every tile shape): the pass is limited by the
                      // memory pipeline, not by how the 3x3 taps are fetched; the simpler form is the default.

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

hipError_t rt_launch_taa_resolve(const void *current, const void *history, const void *normal, void *out, int W, int H,
                                 float blend, float jx, float jy, hipStream_t s) {
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
// Each pass is a pure streaming kernel (8 B/px in + 8 B/px out): 16-byte accesses (two pixels per
// lane), 256-thread workgroups over rows; the vertical pass walks columns through L2.
// =========================================================================================
namespace {

__device__ __forceinline__ unsigned f2h_rtz_u(float f) {
    unsigned u = __float_as_uint(f);
    unsigned s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (a == 0x7f800000u) ? (s | 0x7c00u) : (s | 0x7e00u | ((a >> 13) & 0x1ffu));
    if (a >= 0x47800000u) return s | 0x7bffu;
    if (a >= 0x38800000u) return s | ((a - 0x38000000u) >> 13);
    if (a < 0x33800000u) return s;
    unsigned e = a >> 23, m = (a & 0x7fffffu) | 0x800000u;
    return s | (m >> (126u - e));
}
__device__ __forceinline__ float h2f_u(unsigned h) { return __half2float(__ushort_as_half((unsigned short)(h & 0xffffu))); }
__device__ __forceinline__ uint2 pack_half4(float r, float g, float b) {
    return make_uint2(f2h_rtz_u(r) | (f2h_rtz_u(g) << 16), f2h_rtz_u(b) | (0x3c00u << 16));
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

hipError_t rt_launch_bloom(const void *scene, void *tmpA, void *tmpB, void *out, int W, int H, float threshold, float strength,
                           int iterations, hipStream_t s) {
    const size_t n = (size_t)W * H;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    uint2 *a = (uint2 *)tmpA, *b = (uint2 *)tmpB;
    hipLaunchKernelGGL(rt_bloom_extract_kernel, dim3(blocks), dim3(256), 0, s, (const float4 *)scene, a, n, threshold);
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    bool horizontal = true;                                       // ForwardShadingPipeline.cpp:207
    for (int it = 0; it < iterations; it++) {
        if (horizontal) hipLaunchKernelGGL(rt_bloom_blur_kernel<true>, grid, dim3(256), 0, s, a, b, W, H);
        else hipLaunchKernelGGL(rt_bloom_blur_kernel<false>, grid, dim3(256), 0, s, a, b, W, H);
        uint2 *t = a; a = b; b = t;
        horizontal = !horizontal;
    }
    hipLaunchKernelGGL(rt_bloom_combine_kernel, dim3(blocks), dim3(256), 0, s, (const float4 *)scene, a, (float4 *)out, n, strength);
    return hipGetLastError();
}
