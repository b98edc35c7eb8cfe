// rt_device.h -- shared declarations between the HIP kernels (rt_kernels.hip) and the
// C-ABI implementation (rt_abi.cpp).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_mi355.h"

#define RT_MAX_DEPTH 32      // per-depth bounce-sample table in kernarg memory
#define RT_HALTON_N 64       // tabulated Halton indices (UI range of pcfSamples is 1..16)

// "Compiled" scene record sizes, in float4 units.  See DESIGN.md "Data layout".
#define RT_HOT_F4 6          // per object: AABB + shape, read for every ray
#define RT_MAT_F4 4          // per object: material, read once per closest hit
#define RT_LGT_F4 4          // per light

// Per-launch constants, passed by value in kernarg memory (scalar loads, no staging).
struct RtFrame {
    rt_params p;
    int32_t nObj, nLt;
    int32_t noiseW, noiseH, skySize;
    int32_t anyPcss;                    // some light has shadowType 2: kernel instantiation with paired blocker rays
    int32_t imageStores;                // 1: the surfaces are whole width x height images, a pixel goes to (image row, image column)
                                        // (rt_render_into_image: strips of several devices land in one frame); 0: regionW x regionH window
    float sx, sy;                       // (aspect*tanFov)*focalLength, tanFov*focalLength
    // cosineWeightedHemisphere's local direction for the bounce at each depth
    // (identical for every pixel: hammersley(depth*64+frameCount, 64))
    float hemi[RT_MAX_DEPTH][4];
    float sssHemi[4][4];                // same for computeSubsurfaceScattering's 4 samples
};

struct RtDeviceScene {
    const float4 *compiled;   // [nObj*RT_HOT_F4][nObj*RT_MAT_F4][nLt*RT_LGT_F4][halton2: 16][halton3: 16]
    const uint8_t *noise;     // R8 texels or nullptr
    const uint16_t *sky;      // 6*size*size*3 halfs or nullptr
    // Cost-feedback tile scheduling (packet kernel): workgroup b renders tile tileOrder[b] (or b when
    // null) and records its duration in tileCost[tile]; rt_launch_lpt_sort turns the costs of frame k
    // into the longest-first order of frame k+1.  Pure scheduling: no pixel value depends on it.
    const unsigned *tileOrder;
    unsigned *tileCost;
    // Shadow tables (rt_shadowtab.inc): per light RT_ST_HDR_F4 float4 of header, then the lights' cells (one bit per object);
    // built by rt_set_scene for scenes of <= RT_ST_MAX_OBJECTS objects, nullptr otherwise.
    const unsigned *shadowTab;
};

#define RT_ST_HDR_F4 7            // float4 per light header
#define RT_ST_MAX_OBJECTS 256     // 8 dwords per cell
#define RT_ST_MAX_LIGHTS 64       // (a table is 0.4-6 MB per light)
// dwords per cell for a scene of nObj objects (the packet kernel's profiles are compiled for exactly these: rt_packet.inc)
static inline int rt_shadowtab_words(int nObj) { return nObj <= 32 ? 1 : (nObj <= 64 ? 2 : 8); }
// Table geometry: cube-map cells per face edge (point / area lights), grid cells per axis (directional), bins; and the
// buffer size in dwords that holds any mix of light types.
struct RtShadowTabGeom { int Kcube, Kplan, NB; };
// direction cells of one light (the builder's phase-1 supersets, stored behind the tables)
static inline size_t rt_shadowtab_dir_cells(const RtShadowTabGeom &g) {
    const size_t cube = (size_t)6 * g.Kcube * g.Kcube, plan = (size_t)g.Kplan * g.Kplan;
    return cube > plan ? cube : plan;
}
// dwords of headers + cells (what the render kernel reads, and rt_debug_shadow_tables returns).  `blocker`: the scene has a PCSS
// light -- a second set of tables, for pcssShadow's blocker rays, follows the first (rt_shadowtab.inc)
static inline size_t rt_shadowtab_table_dwords(const RtShadowTabGeom &g, int nObj, int nLt, bool blocker) {
    const size_t cube = (size_t)g.NB * 6 * g.Kcube * g.Kcube, plan = (size_t)g.NB * g.Kplan * g.Kplan + 1;
    return (size_t)nLt * RT_ST_HDR_F4 * 4 + (size_t)(blocker ? 2 : 1) * nLt * (cube > plan ? cube : plan) * rt_shadowtab_words(nObj) + 4;
}
// ... + the builder's scratch
static inline size_t rt_shadowtab_dwords(const RtShadowTabGeom &g, int nObj, int nLt, bool blocker) {
    return rt_shadowtab_table_dwords(g, nObj, nLt, blocker) + (size_t)(blocker ? 2 : 1) * nLt * rt_shadowtab_dir_cells(g) * rt_shadowtab_words(nObj);
}
hipError_t rt_launch_shadow_tables(const float4 *dCompiled, int nObj, int nLt, unsigned *dTab, const RtShadowTabGeom &g, hipStream_t s, bool onePhase = false,
                                   bool blocker = false);

#define RT_PCF_TAB_N 16      // PCF samples tabulated per directional light (UI range of pcfSamples is 1..16)
// float4 count of the whole compiled buffer: the staged part (rt_compiled_f4) + per light RT_PCF_TAB_N x 2 float4
// of precomputed PCF rays (direction, dot(d,d)) (1/direction, -), meaningful for directional lights only
static inline size_t rt_compiled_f4(int nObj, int nLt);
static inline size_t rt_compiled_total_f4(int nObj, int nLt) { return rt_compiled_f4(nObj, nLt) + (size_t)nLt * RT_PCF_TAB_N * 2; }
static inline size_t rt_compiled_f4(int nObj, int nLt) {
    return (size_t)nObj * (RT_HOT_F4 + RT_MAT_F4) + (size_t)nLt * RT_LGT_F4 + 2 * (RT_HALTON_N / 4);
}

// Launch wrappers implemented in rt_kernels.hip
hipError_t rt_launch_compile_scene(const uint8_t *dObjects, int nObj, const uint8_t *dLights, int nLt,
                                   float4 *dCompiled, hipStream_t s);
hipError_t rt_launch_render(const RtFrame &f, const RtDeviceScene &sc, float4 *dColor, float4 *dPos,
                            uint2 *dNormal, unsigned long long *dRayCounter, int variant, hipStream_t s, int countMode = 1);
// Tile geometry of the packet kernel for a given scene size / window (so the ABI layer can size the
// feedback buffers): workgroup threads, tile edge, tiles per row, total tiles.
void rt_packet_geometry(int nObj, int regionW, int regionH, int *bt, int *tile, int *tilesX, int *nTiles);
// Heavy-first tile order predicted from the frame's own inputs (rt_predict_tiles_kernel): no history involved.
hipError_t rt_launch_predict_order(const RtFrame &f, const RtDeviceScene &sc, int tilesX, int nTiles, unsigned *dSeg, int segStride,
                                   unsigned *dCursors, unsigned *dNextCursors, unsigned char *dCls, unsigned *dOrder, hipStream_t s);
hipError_t rt_launch_lpt_sort(unsigned *dCost, unsigned *dSnap, unsigned *dOrder, int nTiles, hipStream_t s, unsigned *dAccum = nullptr);
hipError_t rt_launch_iota(unsigned *dOrder, int n, hipStream_t s);      // dOrder[i] = i (identity tile order)
hipError_t rt_launch_taa_resolve(const void *current, const void *history, const void *normal, void *out, int W, int H,
                                 float blend, float jx, float jy, hipStream_t s);
hipError_t rt_launch_ssao(const void *position, const void *normal, void *depthPlane, void *out, int W, int H, const float *noise,
                          int nW, int nH, const float *samples, const float *projection, const float *view, hipStream_t s);
hipError_t rt_launch_ssao_blur(const void *in, void *out, int W, int H, int horizontal, hipStream_t s);
hipError_t rt_launch_equirect_to_cubemap(const float *dRgb, void *dTex, int W, int H, int S, void *dFaces, hipStream_t s);
hipError_t rt_launch_bloom(const void *scene, void *tmpA, void *tmpB, void *out, int W, int H, float threshold, float strength,
                           int iterations, hipStream_t s);
hipError_t rt_launch_wire_pack(const void *dColor, const void *dPos, const void *dNormal, void *dWire, size_t nPixels,
                               hipStream_t s);
hipError_t rt_launch_wire_unpack(const void *dWire, size_t rankStrideBytes, size_t rankPixels, const void *dRootColor,
                                 const void *dRootPos, const void *dRootNormal, int rootRows, void *dColor, void *dPos,
                                 void *dNormal, int width, int height, int stripRows, int stripCount, hipStream_t s);
hipError_t rt_launch_deinterleave(const void *src, void *dst, int width, int height, int bytesPerPixel,
                                  int stripRows, int stripCount, size_t rankStrideBytes, hipStream_t s);
