// Exhaustive check (all 2^32 float bit patterns, ~1 s on an MI355X) of the correctly-rounded fast paths for 1/x and sqrt(x)
// used by csrc/rt_fastmath.h against the compiler's IEEE sequences (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/fm tools/fastmath_exhaustive.hip && /tmp/fm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define RT_FASTMATH_NO_FALLBACK 1      // test the fast paths alone: report where they need the fallback
#include "../opengl_raytracing_amd/csrc/rt_fastmath.h"
__device__ __forceinline__ bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
// signed zeros distinguished (a quotient's -0 must stay -0)
__device__ __forceinline__ bool same_bits(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
__global__ void check(unsigned long long *out) {
    const unsigned base = (blockIdx.x * 256u + threadIdx.x) * 256u;
    unsigned long long bad[4] = {0, 0, 0, 0};
    for (unsigned k = 0; k < 256u; k++) {
        const float x = __uint_as_float(base + k);
        const unsigned ex = (__float_as_uint(x) >> 23) & 0xffu;
        bool ok1, ok2, ok3;
        const float a = rtf::rcp_fast(x, ok1), s = rtf::sqrt_fast(x, ok2), q = rtf::rcp_sqrt_fast(x, ok3);
        const bool wa = !same(a, 1.0f / x), ws = !same(s, sqrtf(x)), wq = !same(q, 1.0f / sqrtf(x));
        if (wq && ok3) atomicAdd(&out[4 + 514], 1ull);
        if (!same(rtf::rcp_sqrt(x), 1.0f / sqrtf(x))) atomicAdd(&out[4 + 515], 1ull);
        bad[0] += wa && ok1;          // claims to be exact but is not: MUST be 0
        bad[1] += !ok1;               // how many inputs need the IEEE fallback
        bad[2] += ws && ok2;          // MUST be 0
        bad[3] += !ok2;
        if (wa && ok1) atomicAdd(&out[4 + ex], 1ull);
        if (ws && ok2) atomicAdd(&out[4 + 256 + ex], 1ull);
        // the full functions (fast path + wave-level IEEE fallback) must be exact everywhere
        if (!same(rtf::rcp(x), 1.0f / x)) atomicAdd(&out[4 + 512], 1ull);
        if (!same(rtf::sqrt(x), sqrtf(x))) atomicAdd(&out[4 + 513], 1ull);
    }
    for (int i = 0; i < 4; i++) if (bad[i]) atomicAdd(&out[i], bad[i]);
}
// rtf::div2's fast path against the IEEE division on (a, b) pairs: mode 0 = random bit patterns, 1 = random mantissas with
// exponents within +-40 of 1.0 (the perspective divide's range), 2 = structured mantissas (all ones / zeros / +-1 ulp of them)
// with random exponents, 3 = a = RN(q*b) +- few ulps for random q, b (quotients at and next to rounding boundaries),
// 4 = numerators +-0, denormal, 2^-107, 2^-100, 2^123 against denominators of every ordinary exponent.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned structured_mant(unsigned h) {
    switch (h & 7u) { case 0: return 0u; case 1: return 1u; case 2: return 0x7fffffu; case 3: return 0x7ffffeu; case 4: return 0x400000u;
                      case 5: return 0x3fffffu; case 6: return 0x400001u; default: return (h >> 3) & 0x7fffffu; }
}
__global__ void check_div(unsigned long long *out, int mode, unsigned long long seed) {
    const unsigned long long t = ((unsigned long long)blockIdx.x * 256u + threadIdx.x) * 256ull;
    unsigned long long bad = 0, fb = 0, full = 0;
    for (unsigned k = 0; k < 256u; k++) {
        const unsigned long long h = mix64(seed + t + k), h2 = mix64(h);
        unsigned ua = (unsigned)h, ub = (unsigned)(h >> 32), uc = (unsigned)h2;
        if (mode == 1) {
            ua = (ua & 0x807fffffu) | ((87u + ((ua >> 23) & 0xffu) % 81u) << 23);
            ub = (ub & 0x807fffffu) | ((87u + ((ub >> 23) & 0xffu) % 81u) << 23);
            uc = (uc & 0x807fffffu) | ((87u + ((uc >> 23) & 0xffu) % 81u) << 23);
        } else if (mode == 2) {
            ua = (ua & 0xff800000u) | structured_mant((unsigned)(h2 >> 32));
            ub = (ub & 0xff800000u) | structured_mant((unsigned)(h2 >> 43));
            uc = (uc & 0xff800000u) | structured_mant((unsigned)(h2 >> 20));
        }
        float a = __uint_as_float(ua), b = __uint_as_float(ub), c = __uint_as_float(uc);
        if (mode == 4) {                     // signed zeros, tiny and huge numerators against ordinary denominators
            const unsigned sel = (unsigned)(h2 >> 50) % 6u;
            ua = sel == 0 ? 0u : sel == 1 ? 0x80000000u : sel == 2 ? (ua & 0x807fffffu) : sel == 3 ? ((ua & 0x807fffffu) | (20u << 23)) : sel == 4 ? ((ua & 0x807fffffu) | (27u << 23)) : ((ua & 0x807fffffu) | (250u << 23));
            ub = (ub & 0x807fffffu) | ((60u + ((ub >> 23) & 0xffu) % 135u) << 23);
            a = __uint_as_float(ua); b = __uint_as_float(ub);
            c = __uint_as_float(ua ^ 0x80000000u);
        }
        if (mode == 3) {
            ub = (ub & 0x807fffffu) | ((100u + ((ub >> 23) & 0xffu) % 55u) << 23);
            ua = (ua & 0x807fffffu) | ((100u + ((ua >> 23) & 0xffu) % 55u) << 23);
            b = __uint_as_float(ub);
            a = __uint_as_float(__float_as_uint(__uint_as_float(ua) * b) + (int)((h2 >> 40) % 5u) - 2);
            c = __uint_as_float(__float_as_uint(a) ^ 0x1u);
        }
        bool oky, ok0, ok1;
        const float y = rtf::rcp_fast(b, oky);
        const float q0 = rtf::div_fast(a, b, y, ok0), q1 = rtf::div_fast(c, b, y, ok1);
        bad += (oky && ok0 && !same_bits(q0, a / b)) + (oky && ok1 && !same_bits(q1, c / b));
        fb += !(oky && ok0 && ok1);
        float f0, f1;
        rtf::div2(a, c, b, f0, f1);
        full += !same_bits(f0, a / b) + !same_bits(f1, c / b);
    }
    if (bad) atomicAdd(&out[0], bad);
    if (fb) atomicAdd(&out[1], fb);
    if (full) atomicAdd(&out[2], full);
}
static int run_div() {
    unsigned long long *d, h[3];
    (void)hipMalloc(&d, sizeof h);
    int rc = 0;
    for (int mode = 0; mode < 5; mode++) {
        (void)hipMemset(d, 0, sizeof h);
        const int launches = (mode == 1 || mode == 3) ? 2 : 1;
        for (int l = 0; l < launches; l++) check_div<<<65536, 256>>>(d, mode, 0x1234567ull * (mode + 1) + 0x9e3779b97f4a7c15ull * l);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("div2 mode %d: %d x 2^33 quotients: fast path wrong-while-ok %llu (must be 0), pairs needing the fallback %llu, full rtf::div2 mismatches %llu (must be 0)\n",
               mode, launches, h[0], h[1], h[2]);
        rc |= (h[0] || h[2]) ? 1 : 0;
    }
    return rc;
}
int main() {
    unsigned long long *d, h[4 + 516];
    (void)hipMalloc(&d, sizeof h); (void)hipMemset(d, 0, sizeof h);
    check<<<65536, 256>>>(d);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("rcp_fast: wrong-while-ok %llu (must be 0), needs-fallback %llu of 2^32 | sqrt_fast: wrong-while-ok %llu (must be 0), needs-fallback %llu\n", h[0], h[1], h[2], h[3]);
    printf("rcp wrong-while-ok by exponent:"); for (int e = 0; e < 256; e++) if (h[4 + e]) printf(" %d:%llu", e, h[4 + e]);
    printf("\nsqrt wrong-while-ok by exponent:"); for (int e = 0; e < 256; e++) if (h[4 + 256 + e]) printf(" %d:%llu", e, h[4 + 256 + e]);
    printf("\nfull rtf::rcp mismatches %llu, full rtf::sqrt mismatches %llu (both must be 0)\n", h[4 + 512], h[4 + 513]);
    printf("rcp_sqrt_fast wrong-while-ok %llu (must be 0), full rtf::rcp_sqrt mismatches %llu (must be 0)\n", h[4 + 514], h[4 + 515]);
    const int rc_div = run_div();
    return (h[0] || h[2] || h[4 + 512] || h[4 + 513] || h[4 + 514] || h[4 + 515] || rc_div) ? 1 : 0;
}
