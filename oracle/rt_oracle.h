/*
 * rt_oracle.h -- TEST INFRASTRUCTURE.  CPU restatement of the reference ray-tracing
 * compute shader (/root/reference/shader/raytracingCs.glsl:1-584) in scalar fp32 C.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported baseline.  The product
 * (opengl_raytracing_amd/csrc) never includes or links it.
 *
 * Parity status: PINNED.  The restatement is checked against outputs of the reference
 * shader itself, executed unmodified on Mesa llvmpipe by oracle/gl_harness.c
 * (fixtures under tests/golden/, generator tests/golden/make_golden.py).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order/meaning as the uniforms of raytracingCs.glsl:72-89 plus the image
 * size (imageSize(outputImage), :200) and the render window.  Layout is deliberately
 * identical to rt_params in include/rt_mi355.h so one ctypes.Structure serves both. */
typedef struct orc_params {
    float camPos[3], camDir[3], camUp[3], camRight[3];
    float fovDeg;         /* uniform fov (degrees)                     :79 */
    float focalLength;    /* uniform focalLength = 1.0                 :80 */
    float maxRayDistance; /* uniform maxRayDistance = 114514.0         :85 */
    float noiseScale[2];  /* uniform noiseScale                        :88 */
    int32_t frameCount;   /* uniform frameCount                        :89 */
    int32_t useSkybox;    /* uniform useSkybox                         :83 */
    int32_t maxRayDepth;  /* #define MAX_RAY_DEPTH                     :4  */
    int32_t width, height;/* full image size                           :200 */
    /* window in local-row space; output buffers are regionW x regionH */
    int32_t x0, y0, regionW, regionH;
    /* interleaved row strips (multi-GPU tiling): local row ly maps to global row
     * ((ly / stripRows) * stripCount + stripIndex) * stripRows + ly % stripRows.
     * stripCount = 1, stripIndex = 0 is the identity. */
    int32_t stripRows, stripCount, stripIndex;
    int32_t reserved0;    /* DIAGNOSTIC bits, 0 in every parity test against the HIP path: bit 0 = pow(x, 5) by llvmpipe's own
                           * exp2/log2 polynomials instead of the exact product (rt_oracle.c pow5), bit 1 = emulate llvmpipe's
                           * 65 535-iteration loop limiter (rt_oracle.c loop_end).  Both only serve to EXPLAIN what is left
                           * between the restatement and the reference's pixels; oracle/binding.py sets the field explicitly. */
    /* unequal strips: stripCycleRows > 0 -> global row (ly / stripRows) * stripCycleRows +
     * stripOffsetRows + ly % stripRows (stripCount / stripIndex ignored) */
    int32_t stripCycleRows, stripOffsetRows;
} orc_params;

/* Render.  objects: nObj*176 bytes, lights: nLt*96 bytes (std430 layouts,
 * SURVEY.md Appendix B).  noise: noiseW*noiseH R8 texels or NULL.  sky: 6 faces of
 * skySize^2 RGB fp16 (uint16 bits) or NULL.  Outputs: gColor/gPosition regionW*regionH*4
 * floats, gNormal regionW*regionH*4 halfs (RTZ).  rayCount (optional) receives the number
 * of intersectObjects calls.  nthreads <= 0 -> all cores.  Returns 0 on success. */
int orc_render(const void *objects, int nObj, const void *lights, int nLt,
               const orc_params *p, const uint8_t *noise, int noiseW, int noiseH,
               const uint16_t *sky, int skySize, float *gColor, float *gPosition,
               uint16_t *gNormal, uint64_t *rayCount, int nthreads);

/* GenerateAABBForObject (/root/reference/src/SceneIO.h:75-104) applied in place to
 * n 176-byte Object records. */
void orc_generate_aabb(void *objects, int n);

/* Per-function entry points for unit tests (each evaluates n items). */
void orc_halton(const int32_t *index, const int32_t *base, float *out, int n);
void orc_float_to_half_rtz(const float *in, uint16_t *out, int n);
void orc_sample_cube(const uint16_t *sky, int skySize, const float *dirs, float *rgb, int n);

#ifdef __cplusplus
}
#endif
#endif
