// rt_fastmath.h -- correctly-rounded 1/x and sqrt(x) without the compiler's full IEEE sequences.
//
// The reference's arithmetic is IEEE fp32 (x/y and sqrt correctly rounded, normalize(v) =
// v * (1.0/sqrt(dot))), and the parity gate is bit-exact, so every division and square root of the path must be
// correctly rounded.  hipcc's sequences for that handle every exponent: 1.0f/x = v_div_scale x2 + v_rcp + 7 fma/mul +
// v_div_fmas + v_div_fixup (an 11-deep dependent chain, with two denormal-mode switches), sqrtf = ~14 instructions.
// About 40 % of the kernel's VALU instructions were those expansions, and three quarters of the divisions are plain
// reciprocals (ray setup's 1/d per axis, normalize's 1/sqrt).  For operands in the ordinary exponent range much less
// is needed:
//     1/x     : r0 = v_rcp_f32(x) (1 ulp); e = fma(-x, r0, 1); r = fma(e, r0, r0)                  [4 issue slots]
//     sqrt(x) : r = v_rsq_f32(x); g = x*r; h = r/2; d = fma(-g, g, x); s = fma(d, h, g)            [6 issue slots]
// tools/fastmath_exhaustive.hip compares both against the IEEE sequences over ALL 2^32 bit patterns on the GPU:
// the reciprocal is exact whenever its result is a normal number (x = +-0, +-inf and NaN are exact through v_rcp_f32
// itself); the square root is exact for every x >= 2^-100 (below, the residual underflows), x = +-0 passes through.
// Everything else -- denormals, |x| >= 2^126, tiny radicands -- takes the compiler's IEEE sequence through a
// WAVE-UNIFORM branch (one v_cmp_class + one scalar branch on the fast path), so the functions are exact for every
// input, which the same tool verifies.
#pragma once
#include <hip/hip_runtime.h>

namespace rtf {

// v_cmp_class_f32 masks
constexpr int CLS_NAN = 0x003, CLS_NINF = 0x004, CLS_NNORM = 0x008, CLS_NDEN = 0x010, CLS_NZERO = 0x020, CLS_PZERO = 0x040,
              CLS_PDEN = 0x080, CLS_PNORM = 0x100, CLS_PINF = 0x200;

#ifndef RT_FASTMATH_STATS
#define RT_FASTMATH_STATS 0    // 1: count executions of the IEEE fallbacks (diagnostic build, tools/gpu_fastmath_stats.py)
#endif
#if RT_FASTMATH_STATS
__device__ unsigned long long g_fallbacks[4];      // [0] rcp, [1] rcp3, [2] sqrt, [3] rcp_sqrt: wave-level executions
#define RTF_COUNT(i) do { if ((threadIdx.x & 63) == (__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))) atomicAdd(&g_fallbacks[i], 1ull); } while (0)
#else
#define RTF_COUNT(i) do { } while (0)
#endif

// Fast paths; ok = the result is the correctly rounded value.  The guards are the CHEAPEST sufficient ones, not the
// tightest: x = +-0, +-inf, NaN also take the IEEE path (they are rare in live lanes -- measured: not one fallback
// executes on C2..C5, tools/gpu_fastmath_stats.py -- and every instruction on the fast path is paid by every ray).
__device__ __forceinline__ float rcp_fast(float x, bool &ok) {
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    const float r1 = __builtin_fmaf(e, r0, r0);
    ok = __builtin_amdgcn_classf(r1, CLS_NNORM | CLS_PNORM);       // a NORMAL refined result is exact (exhaustive check)
    return r1;
}

__device__ __forceinline__ float sqrt_fast(float x, bool &ok) {
    const float r = __builtin_amdgcn_rsqf(x);
    const float g = x * r, h = 0.5f * r;
    const float d = __builtin_fmaf(-g, g, x);
    ok = (x >= 0x1p-100f) & (x < __builtin_huge_valf());           // below 2^-100 the residual underflows
    return __builtin_fmaf(d, h, g);
}

// 1/sqrt(x) with both roundings (s = RN(sqrt x), then RN(1/s)).  For x in [2^-100, inf) s lies in [2^-50, 2^64), so the
// radicand's range check covers the reciprocal too.  (Seeding the reciprocal with v_rsq_f32's result instead of a
// v_rcp_f32 of s was tried: 228 of 2^32 radicands come out 1 ulp off, with one Newton step or two.)
__device__ __forceinline__ float rcp_sqrt_fast(float x, bool &ok) {
    const float r = __builtin_amdgcn_rsqf(x);
    const float g = x * r, h = 0.5f * r;
    const float d = __builtin_fmaf(-g, g, x);
    const float s = __builtin_fmaf(d, h, g);
    const float r0 = __builtin_amdgcn_rcpf(s);
    const float e = __builtin_fmaf(-s, r0, 1.0f);
    ok = (x >= 0x1p-100f) & (x < __builtin_huge_valf());
    return __builtin_fmaf(e, r0, r0);
}

#ifndef RT_FASTMATH
#define RT_FASTMATH 1
#endif
#ifndef RT_FASTMATH_NOFB
#define RT_FASTMATH_NOFB 0     // 1: drop the fallback branches (NOT exact; timing experiments only)
#endif

// Correctly rounded 1.0f/x for every x.
__device__ __forceinline__ float rcp(float x) {
#if RT_FASTMATH
    bool ok;
    float r = rcp_fast(x, ok);
    if (!RT_FASTMATH_NOFB && __builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0ull, 0)) { RTF_COUNT(0); r = 1.0f / x; }
    return r;
#else
    return 1.0f / x;
#endif
}

// Three at once (ray setup's 1/d per axis): one branch.
__device__ __forceinline__ void rcp3(float x, float y, float z, float &rx, float &ry, float &rz) {
#if RT_FASTMATH
    bool ox, oy, oz;
    rx = rcp_fast(x, ox); ry = rcp_fast(y, oy); rz = rcp_fast(z, oz);
    if (!RT_FASTMATH_NOFB && __builtin_expect(__builtin_amdgcn_ballot_w64(!(ox & oy & oz)) != 0ull, 0)) {
        RTF_COUNT(1);
        rx = 1.0f / x; ry = 1.0f / y; rz = 1.0f / z;
    }
#else
    rx = 1.0f / x; ry = 1.0f / y; rz = 1.0f / z;
#endif
}

// Correctly rounded sqrtf(x) for every x.
__device__ __forceinline__ float sqrt(float x) {
#if RT_FASTMATH
    bool ok;
    float s = sqrt_fast(x, ok);
    if (!RT_FASTMATH_NOFB && __builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0ull, 0)) { RTF_COUNT(2); s = sqrtf(x); }
    return s;
#else
    return sqrtf(x);
#endif
}

// 1.0f / sqrtf(x), both steps correctly rounded (normalize(v) = v * (1.0/sqrt(dot(v,v)))).
__device__ __forceinline__ float rcp_sqrt(float x) {
#if RT_FASTMATH
    bool ok;
    float r = rcp_sqrt_fast(x, ok);
    if (!RT_FASTMATH_NOFB && __builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0ull, 0)) { RTF_COUNT(3); r = 1.0f / sqrtf(x); }
    return r;
#else
    return 1.0f / sqrtf(x);
#endif
}

// Correctly rounded quotients a/b that SHARE a reciprocal (a perspective divide, vec3 / float, a per-ray denominator).
// With y = RN(1/b) from rcp_fast above: q = a*y is within an ulp of a/b, r = fma(-b, q, a) is the exact remainder, and
// RN(q + r*y) is the correctly rounded quotient (Markstein's theorem) -- as long as nothing under- or overflows on the
// way: y and the result normal numbers, |a| >= 2^-100 (below, the remainder's low bits fall under 2^-149).  a = +-0 is
// exact as q = a*y itself (the corrected value would lose the sign of -0).  3 issue slots + 3 compares per quotient and 3
// per reciprocal, against 11 with two mode switches for each IEEE division; callers OR the `ok` flags of a whole group of
// divisions and redo the group with the IEEE sequences in a wave-uniform branch when any is false.
// tools/fastmath_exhaustive.hip: 6 x 2^33 random, structured and boundary (a, b) pairs, no mismatch while ok.
__device__ __forceinline__ float div_fast(float a, float b, float y, bool &ok) {
    const float q = a * y;
    const float r = __builtin_fmaf(-b, q, a);
    const float q1 = __builtin_fmaf(r, y, q);
    const bool zero = a == 0.0f;
    ok = (__builtin_amdgcn_classf(q1, CLS_NNORM | CLS_PNORM) & (__builtin_fabsf(a) >= 0x1p-100f)) | zero;
    return zero ? q : q1;
}

// vec3 / float (one reciprocal, three quotients), correctly rounded for every input.
__device__ __forceinline__ void div3(float a0, float a1, float a2, float b, float &q0, float &q1, float &q2) {
#if RT_FASTMATH
    bool oky, ok0, ok1, ok2;
    const float y = rcp_fast(b, oky);
    q0 = div_fast(a0, b, y, ok0);
    q1 = div_fast(a1, b, y, ok1);
    q2 = div_fast(a2, b, y, ok2);
    if (!RT_FASTMATH_NOFB && __builtin_expect(__builtin_amdgcn_ballot_w64(!(oky & ok0 & ok1 & ok2)) != 0ull, 0)) { q0 = a0 / b; q1 = a1 / b; q2 = a2 / b; }
#else
    q0 = a0 / b; q1 = a1 / b; q2 = a2 / b;
#endif
}

__device__ __forceinline__ void div2(float a0, float a1, float b, float &q0, float &q1) {
#if RT_FASTMATH
    bool oky, ok0, ok1;
    const float y = rcp_fast(b, oky);
    q0 = div_fast(a0, b, y, ok0);
    q1 = div_fast(a1, b, y, ok1);
    if (!RT_FASTMATH_NOFB && __builtin_expect(__builtin_amdgcn_ballot_w64(!(oky & ok0 & ok1)) != 0ull, 0)) { q0 = a0 / b; q1 = a1 / b; }
#else
    q0 = a0 / b; q1 = a1 / b;
#endif
}

}  // namespace rtf
