// Exhaustive check (all 2^32 float bit patterns, ~1 s on an MI355X) of the correctly-rounded fast paths for 1/x and sqrt(x)
// used by csrc/rt_fastmath.h against the compiler's IEEE sequences (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/fm tools/fastmath_exhaustive.hip && /tmp/fm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define RT_FASTMATH_NO_FALLBACK 1      // test the fast paths alone: report where they need the fallback
#include "../opengl_raytracing_amd/csrc/rt_fastmath.h"
__device__ __forceinline__ bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
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
    return (h[0] || h[2] || h[4 + 512] || h[4 + 513] || h[4 + 514] || h[4 + 515]) ? 1 : 0;
}
