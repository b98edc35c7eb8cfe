// rt_mesa_math.h -- the reference GL's own sin / cos / tan / exp, restated for host and device.
//
// The reference shader is written against a GL driver; its golden pixels come from Mesa llvmpipe.  Three places of
// the path take a transcendental of a value that then steers DISCRETE decisions or every pixel at once:
//   tan(radians(fov)*0.5)                (raytracingCs.glsl:209)  -> scale of every camera ray        [host]
//   cos/sin(2*PI*rand.x)                 (:296-298)               -> the bounce sample all pixels share [host]
//   sin(dot(st, (12.9898, 78.233)))      (:274, random())         -> Russian-roulette decisions        [device]
// and exp(-t/scatterDistance) (:334) scales the subsurface term [device].  Any 1-ulp deviation from the driver's
// polynomial shows up as silhouette / roulette flips against the reference's pixels, so these four are evaluated
// exactly as Mesa 23.2 / gallivm does (tan = sin/cos by the GLSL front end; lp_build_sin_or_cos = the cephes /
// sse_mathfun single-precision sincos with llvm.fmuladd in the reduction and the polynomials; exp(x) =
// exp2(x*log2 e) with lp_build_exp2's degree-5 polynomial by the even/odd Horner split) -- restated from the
// published algorithm, NOT libm.  The CPU oracle carries its own restatement (oracle/rt_oracle.c), which is pinned
// bitwise against llvmpipe (tests/golden/trig.npz); the GPU tests pin this one against the oracle's through the
// rendered pixels (bit-exact surfaces on every config) and tests/test_abi_host.py::test_mesa_trig_host.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace rtm {

__host__ __device__ inline float bits_f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__host__ __device__ inline uint32_t f_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

// want_cos = 0: sin(a), 1: cos(a).  Exact (vs llvmpipe) for |a| < 1.6e9 and for inf / NaN.
__host__ __device__ inline float sincos(float a, int want_cos) {
    const uint32_t ai = f_bits(a);
    const float x = bits_f(ai & 0x7fffffffu);
    const float y = x * 1.27323954473516f;                                  // 4/pi
    const int32_t j = (fabsf(y) < 2147483648.0f) ? (int32_t)y : INT32_MIN;  // x86 cvttps2dq semantics
    const int32_t jadd = (int32_t)((uint32_t)j + 1u);
    const int32_t jand = jadd & ~1;
    const float y2 = (float)jand;
    const int32_t e2 = want_cos ? (int32_t)((uint32_t)jand - 2u) : jand;
    const uint32_t sign = want_cos ? (((uint32_t)(4 & ~e2)) << 29) : ((ai ^ ((uint32_t)jadd << 29)) & 0x80000000u);
    float r = __builtin_fmaf(y2, -0.78515625f, x);                          // Cody-Waite, three constants, fused
    r = __builtin_fmaf(y2, -2.4187564849853515625e-4f, r);
    r = __builtin_fmaf(y2, -3.77489497744594108e-8f, r);
    const float z = r * r;
    float c = __builtin_fmaf(z, 2.443315711809948E-005f, -1.388731625493765E-003f);
    c = __builtin_fmaf(c, z, 4.166664568298827E-002f);
    c = (c * z) * z;
    c = (c - z * 0.5f) + 1.0f;
    float s = __builtin_fmaf(z, -1.9515295891E-4f, 8.3321608736E-3f);
    s = __builtin_fmaf(s, z, -1.6666654611E-1f);
    s = __builtin_fmaf(s * z, r, r);
    float v = ((e2 & 2) == 0) ? s : c;
    v = bits_f(f_bits(v) ^ sign);
    v = fminf(fmaxf(v, -1.0f), 1.0f);
    if (!(fabsf(a) < __builtin_huge_valf())) v = bits_f(0x7fc00000u);
    return v;
}
__host__ __device__ inline float sin_(float a) { return sincos(a, 0); }
__host__ __device__ inline float cos_(float a) { return sincos(a, 1); }
__host__ __device__ inline float tan_(float a) { return sincos(a, 0) / sincos(a, 1); }

// exp(x) = exp2(x * fl(log2 e)); exp2 = 2^floor(x) * Q(x - floor(x)), x clamped to [-126.99999, 128]
__host__ __device__ inline float exp_(float x0) {
    float x = x0 * 1.44269504088896340736f;
    if (x != x) return x;
    x = fmaxf(-126.99999f, fminf(128.0f, x));
    const float ip = floorf(x), fp = x - ip;
    const float e = bits_f((uint32_t)((int32_t)ip + 127) << 23);
    const float f2 = fp * fp;
    // lp_build_polynomial: even and odd coefficients by separate Horner chains in fp^2, joined by one fused step
    float even = __builtin_fmaf(f2, 0.00898934009049466391101f, 0.240153617044375388211f);
    even = __builtin_fmaf(f2, even, 1.0f);
    float odd = __builtin_fmaf(f2, 0.00187757667519147912699f, 0.0558263180532956664775f);
    odd = __builtin_fmaf(f2, odd, 0.693153073200168932794f);
    return e * __builtin_fmaf(odd, fp, even);
}

}  // namespace rtm
