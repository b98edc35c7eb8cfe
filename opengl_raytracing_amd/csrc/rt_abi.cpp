// rt_abi.cpp -- implementation of the C ABI declared in include/rt_mi355.h: context,
// device buffers, per-frame constants, launches, readback, timing.  Each entry point
// stands in for a piece of the reference's dispatch site
// (/root/reference/src/ForwardShadingPipeline.cpp:155-182); see the header for the map.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "rt_device.h"
#include "rt_mesa_math.h"

struct rt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t evStart = nullptr, evStop = nullptr, evScene = nullptr;
    bool timed = false;
    // raw SSBO bytes + compiled scene
    uint8_t *dObjects = nullptr, *dLights = nullptr;
    size_t capObjects = 0, capLights = 0;
    float4 *dCompiled = nullptr;
    size_t capCompiledF4 = 0;
    int nObj = 0, nLt = 0;
    bool anyPcss = false;               // some light of the current scene has shadowType 2 (selects the kernel instantiation)
    // double-buffered pinned staging so rt_set_scene never blocks on the GPU and the caller's
    // bytes are consumed before it returns (glBufferData semantics)
    uint8_t *hStage[2] = {nullptr, nullptr};
    size_t capStage[2] = {0, 0};
    hipEvent_t evStage[2] = {nullptr, nullptr};
    bool stageUsed[2] = {false, false};
    unsigned stageSeq = 0;
    // textures
    uint8_t *dNoise = nullptr;
    int noiseW = 0, noiseH = 0;
    uint16_t *dSky = nullptr;
    int skySize = 0;
    // context-owned output surfaces
    float4 *dColor = nullptr, *dPos = nullptr;
    uint2 *dNormal = nullptr;
    size_t capPixels = 0;
    int surfW = 0, surfH = 0;
    unsigned long long *dRayCounter = nullptr;
    int variant = 1;   // 1 = wavefront-packet kernel (default), 0 = exhaustive per-lane loop
    unsigned long long lastStats[32] = {};   // rt_count_rays diagnostics (rt_debug_stats)
    // cost-feedback tile scheduling state (packet kernel)
    // Frames may be issued on several streams (frames in flight overlapping on the device), so the order is
    // double-buffered: a sort writes the buffer no launch is reading, and later launches wait for it.
    unsigned *dTileCost = nullptr, *dTileSnap = nullptr, *dTileOrder[2] = {nullptr, nullptr};
    int fbCur = -1;                            // order buffer new launches read (-1 = raster order)
    int fbNext = 0;                            // order buffer the pending sort is writing
    bool sortPending = false;                  // a sort has been enqueued on the context's own stream, not yet adopted
    unsigned sortAge = 0;                      // fbAge at which it was enqueued
    size_t capTiles = 0;
    int fbTiles = 0, fbTilesX = 0, fbBt = 0;   // geometry the current order was measured on (0 = none)
    unsigned fbAge = 0;                        // frames since that geometry was first seen
    // Every stream a render has been issued on (the context's own and the callers'): `last` = its most recent
    // launch (recorded on EVERY launch, feedback or not: rt_set_scene orders the scene rewrite behind all of
    // them), `seenGen` = the tile-order adoption it has already ordered itself behind.
    struct FbStream {
        hipStream_t s; hipEvent_t last; unsigned seenGen;
        // predicted tile order of the last frame issued on this stream, and the inputs it was predicted from
        // (class segments: 32 x predCap tile ids; two sets of 32 class sizes, used alternately -- a prediction clears the other set)
        unsigned *predOrder; unsigned char *predCls; unsigned *predCounts; size_t predCap; unsigned long long predKey; bool predValid; int predSet;
    };
    static constexpr int kMaxStreams = 8;
    FbStream fbStreams[kMaxStreams] = {};
    int nFbStreams = 0;
    hipEvent_t evSort[2] = {nullptr, nullptr}; // completion of the rt_lpt_sort that wrote dTileOrder[k]
    unsigned adoptGen = 0;                     // bumped whenever fbCur changes to a freshly sorted buffer
    void *dBloom[2] = {nullptr, nullptr};      // rgba16f ping-pong targets of rt_bloom
    float *dSsaoDepth = nullptr;               // gPosition.z plane of rt_ssao
    size_t capSsaoPx = 0;
    // rt_frame: AO result (raw, blurred), TAA history ping-pong, bloom-combined image when the caller passes none
    float *dFrameAO[2] = {nullptr, nullptr};
    float4 *dHistory[2] = {nullptr, nullptr};
    float4 *dFrameDisplay = nullptr;
    size_t capFramePx = 0;
    int frameW = 0, frameH = 0, lastHistory = -1;
    bool frameAOValid = false;
    size_t capBloomPx = 0;
    // shadow tables of the current scene (rt_shadowtab.inc)
    unsigned *dShadowTab = nullptr;
    size_t capShadowTab = 0;
    bool shadowTabValid = false;
    bool shadowTabBlocker = false;             // the buffer also holds the blocker-ray tables of the scene's PCSS lights
    std::vector<uint8_t> lastScene;            // the bytes of the current scene (objects, then lights): an identical re-upload is a no-op
    bool stOnePhase = false;                   // RT_ST_BUILD=full: the one-phase builder (every object in every cell; comparison builds only)
    RtShadowTabGeom stGeomSmall = {48, 96, 32}, stGeomLarge = {32, 64, 32};      // <= 32 objects / more (RT_ST_GEOM overrides both); measured: DESIGN.md
    // Scheduler (packet kernel).  schedMode 2 (default): tiles run longest-first by their MEASURED cost over the last frames of the
    // same window geometry (the feedback of rounds 1-2); while no measured order exists yet, frames of >= predMinTiles tiles run in
    // the heavy-first order PREDICTED from their own inputs (rt_predict_tiles_kernel, in the frame's own stream, buffers per stream
    // in FbStream).  1: measured costs only.  0: raster order.
    int schedMode = 2;
    int predMinTiles = 49152;                  // frames of at least this many tiles get a predicted order while no measured one exists (RT_PRED_MIN_TILES)
    unsigned sceneGen = 0, texGen = 0;         // bumped by rt_set_scene / rt_set_noise / rt_set_skybox / rt_equirect_to_cubemap
    bool dbgWantCls = false;                   // RT_DEBUG_PRED_CLASSES=1: predictions also store each tile's class
    const unsigned char *dbgPredCls = nullptr; // class buffer of the last predicted launch (rt_debug_predicted_classes)
    int dbgPredTiles = 0;
    bool feedback = true;
    unsigned fbPeriod = 32;                    // re-sort period in frames (RT_FB_PERIOD overrides, for measurements)
    // Free-running frameCount (the reference with TAA on, ForwardShadingPipeline.cpp:254): frameCount enters the frame through
    // hammersley(depth*64 + frameCount, 64) (:557) -- the bounce sample ALL pixels share -- whose azimuth is periodic in frameCount
    // with period 64 and whose cos^2(theta) = halton2 repeats to within 2^-6.  A frame's tile costs therefore repeat, nearly, every
    // 64 frames, while consecutive frames differ a lot (tools/gpu_phase_costs.py: list-scheduling makespan over the ideal, C2:
    // 1.04 in the frame's own order, 1.06-1.09 in the order of the frame 64 earlier, 1.07-1.17 in the all-phase average order,
    // 1.29-1.34 in raster order).  So once frameCount is seen advancing, every frame's costs go to its PHASE's buffer
    // (frameCount mod 64) and are sorted, beside the following frames on a stream of their own, into that phase's order, which
    // the frame 64 later runs in; phases not seen yet use the all-phase average order as before (the per-phase sorts feed it).
    hipStream_t phaseStream = nullptr;
    unsigned *dPhaseCost = nullptr, *dPhaseOrder = nullptr, *dPhaseSnap = nullptr;   // [64][phaseTiles], [64][phaseTiles], [phaseTiles]
    size_t capPhase = 0;                       // tiles per phase the buffers hold
    int phaseTiles = 0;                        // tiles per phase of the current geometry (0 = none)
    hipEvent_t evPhase[64] = {};               // completion of the last sort into phase k's order
    unsigned char phaseState[64] = {};         // 0 = no order, 1 = sort issued, 2 = seen complete
    bool phaseOn = true;                       // RT_PHASE_ORDER=0 switches it off (measurements)
    bool haveLastFc = false;
    int lastFc = 0, freeRun = 0;               // consecutive scheduled launches whose frameCount differed from the previous one's
    std::string err;
};

namespace {

int fail(rt_context *c, int code, const char *what, hipError_t e = hipSuccess) {
    if (c) {
        c->err = what;
        if (e != hipSuccess) {
            c->err += ": ";
            c->err += hipGetErrorString(e);
        }
    }
    return code;
}

#define HIP_TRY(c, call)                                        \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return fail(c, RT_ERR_HIP, #call, e_); \
    } while (0)

template <typename T>
int ensure(rt_context *c, T **p, size_t *cap, size_t need) {
    if (need <= *cap && *p) return RT_OK;
    if (*p) HIP_TRY(c, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    size_t n = need ? need : 1;
    HIP_TRY(c, hipMalloc((void **)p, n * sizeof(T)));
    *cap = n;
    return RT_OK;
}

// haltonSequence (/root/reference/shader/raytracingCs.glsl:278-288), host copy for the
// per-depth bounce sample.
float halton_host(int index, int base) {
    float result = 0.0f;
    float f = 1.0f / (float)base;
    int i = index;
    while (i > 0) {
        result += f * (float)(i % base);
        i = i / base;
        f = f / (float)base;
    }
    return result;
}

// cosineWeightedHemisphere's local direction (:292-300) for a given (rand.x, rand.y)
void hemi_local(float rx, float ry, float out[4]) {
    const float PI_F = 3.14159265359f;
    float phi = 2.0f * PI_F * rx;
    float cosTheta = sqrtf(ry);
    float sinTheta = sqrtf(1.0f - ry);
    out[0] = sinTheta * rtm::cos_(phi);      // the reference GL's own cos / sin (rt_mesa_math.h), not libm's
    out[1] = cosTheta;
    out[2] = sinTheta * rtm::sin_(phi);
    out[3] = 0.0f;
}

int validate_params(rt_context *c, const rt_params *p) {
    if (!p) return fail(c, RT_ERR_INVALID_ARG, "params is NULL");
    if (p->width <= 0 || p->height <= 0) return fail(c, RT_ERR_INVALID_ARG, "width/height must be positive");
    if (p->regionW < 0 || p->regionH < 0 || p->x0 < 0 || p->y0 < 0)
        return fail(c, RT_ERR_INVALID_ARG, "negative window");
    if (p->stripRows <= 0) return fail(c, RT_ERR_INVALID_ARG, "bad strip mapping");
    if (p->stripCycleRows > 0) {
        if (p->stripOffsetRows < 0 || p->stripOffsetRows + p->stripRows > p->stripCycleRows)
            return fail(c, RT_ERR_INVALID_ARG, "bad strip mapping (offset + rows exceed the cycle)");
    } else if (p->stripCycleRows < 0 || p->stripCount <= 0 || p->stripIndex < 0 || p->stripIndex >= p->stripCount) {
        return fail(c, RT_ERR_INVALID_ARG, "bad strip mapping");
    }
    if (p->maxRayDepth < 0 || p->maxRayDepth > RT_MAX_DEPTH)
        return fail(c, RT_ERR_INVALID_ARG, "maxRayDepth outside [0, 32]");
    return RT_OK;
}

void build_frame(const rt_context *c, const rt_params *p, RtFrame *f) {
    memset(f, 0, sizeof *f);
    f->p = *p;
    if (p->stripCycleRows <= 0) {      // equal strips: normalise to the cycle/offset form the kernels use
        f->p.stripCycleRows = p->stripRows * p->stripCount;
        f->p.stripOffsetRows = p->stripIndex * p->stripRows;
    }
    f->nObj = c->nObj;
    f->nLt = c->nLt;
    f->anyPcss = c->anyPcss ? 1 : 0;
    f->noiseW = c->noiseW;
    f->noiseH = c->noiseH;
    f->skySize = c->skySize;
    // generateCameraRay (:208-211): aspect, tan(radians(fov)*0.5); radians(x) = x*fl(pi/180)
    float aspect = (float)p->width / (float)p->height;
    // Mesa's tan = sin/cos by its own polynomials (rt_mesa_math.h); libm's tanf is 1 ulp off at fov 45, which
    // perturbs every camera ray and showed up as silhouette flips against the reference's pixels
    float tanFov = rtm::tan_((p->fovDeg * 0.017453292519943295f) * 0.5f);
    f->sx = aspect * tanFov * p->focalLength;
    f->sy = tanFov * p->focalLength;
    // hammersley(depth*64 + frameCount, 64) (:557, :311-313): same sample for every pixel
    for (int d = 0; d < RT_MAX_DEPTH; d++) {
        int hi = d * 64 + p->frameCount;
        hemi_local((float)hi / 64.0f, halton_host(hi, 2), f->hemi[d]);
    }
    for (int i = 0; i < 4; i++) hemi_local((float)i / 4.0f, halton_host(i, 2), f->sssHemi[i]);   // :320
}

hipError_t fb_sync_all(rt_context *c) {
    hipError_t e = hipStreamSynchronize(c->stream);
    if (c->phaseStream && e == hipSuccess) e = hipStreamSynchronize(c->phaseStream);
    for (int i = 0; i < c->nFbStreams && e == hipSuccess; i++) e = hipEventSynchronize(c->fbStreams[i].last);
    for (int k = 0; k < 2 && e == hipSuccess; k++)
        if (c->evSort[k]) e = hipEventSynchronize(c->evSort[k]);
    return e;
}

// The record of stream s (created on first use).  More streams than slots: drain and start over (not a hot path).
int stream_record(rt_context *c, hipStream_t s, rt_context::FbStream **out) {
    for (int i = 0; i < c->nFbStreams; i++)
        if (c->fbStreams[i].s == s) { *out = &c->fbStreams[i]; return RT_OK; }
    if (c->nFbStreams == rt_context::kMaxStreams) {
        hipError_t e = fb_sync_all(c);
        if (e != hipSuccess) return fail(c, RT_ERR_HIP, "fb_sync_all", e);
        c->nFbStreams = 0;
    }
    rt_context::FbStream *m = &c->fbStreams[c->nFbStreams++];
    m->s = s;
    m->predValid = false;
    m->seenGen = 0;                // adoptGen starts at 1 with the first adoption: a new stream always orders itself
    if (!m->last) {
        hipError_t e = hipEventCreateWithFlags(&m->last, hipEventDisableTiming);
        if (e != hipSuccess) { c->nFbStreams--; return fail(c, RT_ERR_HIP, "hipEventCreateWithFlags", e); }
    }
    hipError_t e = hipEventRecord(m->last, s);
    if (e != hipSuccess) return fail(c, RT_ERR_HIP, "hipEventRecord", e);
    *out = m;
    return RT_OK;
}

hipError_t fb_wait_others(rt_context *c, const rt_context::FbStream *mine, hipStream_t s) {
    for (int i = 0; i < c->nFbStreams; i++) {
        if (&c->fbStreams[i] == mine) continue;
        hipError_t e = hipStreamWaitEvent(s, c->fbStreams[i].last, 0);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

int launch(rt_context *c, const rt_params *p, float4 *dColor, float4 *dPos, uint2 *dNormal,
           unsigned long long *counter, hipStream_t s, bool timed, int countMode = 1, bool imageStores = false) {
    RtFrame f;
    build_frame(c, p, &f);
    f.imageStores = imageStores ? 1 : 0;
    RtDeviceScene sc;
    sc.compiled = c->dCompiled;
    sc.noise = c->dNoise;
    sc.sky = c->dSky;
    sc.tileOrder = nullptr;
    sc.tileCost = nullptr;
    sc.shadowTab = c->shadowTabValid ? c->dShadowTab : nullptr;
    if (!c->dCompiled) return fail(c, RT_ERR_INVALID_ARG, "rt_set_scene has not been called");
    if (c->variant != 1 && (c->nObj > RT_EXHAUSTIVE_MAX_OBJECTS || c->nLt > RT_EXHAUSTIVE_MAX_LIGHTS))
        return fail(c, RT_ERR_TOO_LARGE, "the exhaustive cross-check kernel stages the scene in LDS: at most 512 objects / 64 lights (use the default kernel)");
    // Longest-first tile order from the previous frames' measured tile costs (same window geometry, any
    // stream).  The first frame of a geometry runs in raster order and only records costs.
    bool sortAfter = false;
    int bt = 0, tile = 0, tilesX = 0, nTiles = 0;
    rt_context::FbStream *mine = nullptr;
    {
        const int rc = stream_record(c, s, &mine);
        if (rc) return rc;
    }
    const bool sched = c->variant == 1 && c->schedMode != 0 && c->feedback && !counter && p->regionW > 0 && p->regionH > 0;
    if (sched) {
        rt_packet_geometry(c->nObj, p->regionW, p->regionH, &bt, &tile, &tilesX, &nTiles);
        if ((size_t)nTiles > c->capTiles) {
            HIP_TRY(c, fb_sync_all(c));
            for (unsigned **q : {&c->dTileCost, &c->dTileSnap, &c->dTileOrder[0], &c->dTileOrder[1]}) {
                if (*q) HIP_TRY(c, hipFree(*q));
                *q = nullptr;
            }
            c->capTiles = 0;
            c->fbTiles = 0;
            c->fbCur = -1;
            c->sortPending = false;
            HIP_TRY(c, hipMalloc((void **)&c->dTileCost, (size_t)nTiles * sizeof(unsigned)));
            HIP_TRY(c, hipMalloc((void **)&c->dTileSnap, (size_t)nTiles * sizeof(unsigned)));
            HIP_TRY(c, hipMalloc((void **)&c->dTileOrder[0], (size_t)nTiles * sizeof(unsigned)));
            HIP_TRY(c, hipMalloc((void **)&c->dTileOrder[1], (size_t)nTiles * sizeof(unsigned)));
            c->capTiles = (size_t)nTiles;
        }
        // the inputs this frame's tile costs are a function of
        unsigned long long key = 1469598103934665603ull;
        {
            const unsigned char *fb = (const unsigned char *)&f;
            for (size_t k = 0; k < sizeof f; k++) key = (key ^ fb[k]) * 1099511628211ull;
            const unsigned gens[3] = {c->sceneGen, c->texGen, (unsigned)c->variant};
            const unsigned char *gb = (const unsigned char *)gens;
            for (size_t k = 0; k < sizeof gens; k++) key = (key ^ gb[k]) * 1099511628211ull;
        }
        const bool same = c->fbTiles == nTiles && c->fbTilesX == tilesX && c->fbBt == bt;
        if (!same) {
            // New geometry: nobody may still be using the old costs / orders, and a sort of the old geometry must
            // never be adopted.  Both order buffers restart as the identity permutation (a launch that reads one
            // before its first sort -- it cannot, fbCur is -1 -- would still visit every tile exactly once).
            HIP_TRY(c, fb_wait_others(c, mine, s));
            for (int k = 0; k < 2; k++)
                if (c->evSort[k]) HIP_TRY(c, hipStreamWaitEvent(s, c->evSort[k], 0));
            HIP_TRY(c, hipMemsetAsync(c->dTileCost, 0, (size_t)nTiles * sizeof(unsigned), s));
            HIP_TRY(c, rt_launch_iota(c->dTileOrder[0], nTiles, s));
            HIP_TRY(c, rt_launch_iota(c->dTileOrder[1], nTiles, s));
            HIP_TRY(c, hipEventRecord(mine->last, s));      // later sorts (context stream) order behind the re-initialisation
            c->fbCur = -1;
            c->sortPending = false;
            c->fbAge = 0;
            // the phases' orders belong to the old geometry.  Sorts still queued on the phase stream finish before any
            // sort of the new geometry (one stream), and an order is read only once its own sort has been seen complete.
            memset(c->phaseState, 0, sizeof c->phaseState);
            c->phaseTiles = 0;
        } else if (c->sortPending) {
            // The sort runs on the context's own stream, beside the frames; its order is adopted by the first launch
            // issued after it has completed, so no render stream waits for a sort in steady state.
            const hipError_t q = hipEventQuery(c->evSort[c->fbNext]);
            (void)hipGetLastError();           // hipErrorNotReady is an answer, not an error to find later
            // Not ready: the host is running ahead of the device.  No order at all yet (second frame of a geometry),
            // or the sort was issued more than one launch per render stream ago (it sits behind a frame that is no
            // longer in flight when this launch reaches the device, so it is all but done): adopt it anyway -- every
            // stream orders itself behind the sort's event on its next launch (seenGen below).
            if (q == hipSuccess || c->fbCur < 0 || c->fbAge - c->sortAge >= 1u + (unsigned)c->nFbStreams) {
                c->fbCur = c->fbNext;
                c->sortPending = false;
                c->adoptGen++;
            }
        }
        // EVERY stream's first launch after an adoption waits for the sort that wrote the adopted buffer (a no-op
        // once it has completed).  Adoption state is context-global, streams are not ordered with each other: without
        // this, a frame on stream B could read an order that only stream A had waited for.
        if (c->fbCur >= 0 && mine->seenGen != c->adoptGen) {
            HIP_TRY(c, hipStreamWaitEvent(s, c->evSort[c->fbCur], 0));
            mine->seenGen = c->adoptGen;
        }
        sc.tileOrder = c->fbCur >= 0 ? c->dTileOrder[c->fbCur] : nullptr;
        sc.tileCost = c->dTileCost;
        sortAfter = true;
        // No measured order yet (the first frames of a geometry): the order PREDICTED from the frame's own inputs, made in this stream
        // just before the frame (cached per stream and inputs).  Only for frames large enough that the pass (one path per tile against
        // every object, 36 us at 1080p / 18 objects) costs less than the order gains: measured it pays from 4K on
        // (C4: 6.97 -> 6.61 ms), at 1080p the raster order of the configs' scenes is already close to heavy-first and it does not.
        const double work = (double)nTiles * (double)(p->maxRayDepth > 0 ? p->maxRayDepth : 1) * (double)(c->nObj > 0 ? c->nObj : 1);
        if (c->fbCur < 0 && c->schedMode == 2 && nTiles >= c->predMinTiles && c->nObj <= RT_ST_MAX_OBJECTS && work <= 3.0e8) {
            if ((size_t)nTiles > mine->predCap) {
                HIP_TRY(c, hipStreamSynchronize(s));
                if (mine->predOrder) HIP_TRY(c, hipFree(mine->predOrder));
                if (mine->predCls) HIP_TRY(c, hipFree(mine->predCls));
                mine->predOrder = nullptr; mine->predCls = nullptr; mine->predCap = 0; mine->predValid = false;
                HIP_TRY(c, hipMalloc((void **)&mine->predOrder, (size_t)33 * nTiles * sizeof(unsigned)));      // the order, then the 32 class segments
                HIP_TRY(c, hipMalloc((void **)&mine->predCls, (size_t)nTiles));
                if (!mine->predCounts) {
                    HIP_TRY(c, hipMalloc((void **)&mine->predCounts, 64 * sizeof(unsigned)));
                    HIP_TRY(c, hipMemsetAsync(mine->predCounts, 0, 64 * sizeof(unsigned), s));
                    mine->predSet = 1;
                }
                mine->predCap = (size_t)nTiles;
            }
            if (!mine->predValid || mine->predKey != key) {
                mine->predSet ^= 1;
                HIP_TRY(c, rt_launch_predict_order(f, sc, tilesX, nTiles, mine->predOrder + mine->predCap, (int)mine->predCap,
                                                   mine->predCounts + 32 * mine->predSet, mine->predCounts + 32 * (mine->predSet ^ 1),
                                                   c->dbgWantCls ? mine->predCls : nullptr, mine->predOrder, s));
                mine->predKey = key;
                mine->predValid = true;
            }
            sc.tileOrder = mine->predOrder;
            c->dbgPredCls = mine->predCls;
            c->dbgPredTiles = nTiles;
        }
    }
    int phase = -1;
    if (sched) {
        c->freeRun = (c->haveLastFc && p->frameCount != c->lastFc) ? (c->freeRun < 1000 ? c->freeRun + 1 : 1000) : 0;
        c->lastFc = p->frameCount;
        c->haveLastFc = true;
        if (c->phaseOn && c->schedMode == 2 && c->freeRun >= 2 && (size_t)nTiles * 64 * 2 * sizeof(unsigned) <= ((size_t)1 << 30)) {
            if (!c->phaseStream) HIP_TRY(c, hipStreamCreateWithFlags(&c->phaseStream, hipStreamNonBlocking));
            if ((size_t)nTiles > c->capPhase) {
                HIP_TRY(c, fb_sync_all(c));
                for (unsigned **q : {&c->dPhaseCost, &c->dPhaseOrder, &c->dPhaseSnap}) {
                    if (*q) HIP_TRY(c, hipFree(*q));
                    *q = nullptr;
                }
                c->capPhase = 0;
                HIP_TRY(c, hipMalloc((void **)&c->dPhaseCost, (size_t)nTiles * 64 * sizeof(unsigned)));
                HIP_TRY(c, hipMalloc((void **)&c->dPhaseOrder, (size_t)nTiles * 64 * sizeof(unsigned)));
                HIP_TRY(c, hipMalloc((void **)&c->dPhaseSnap, (size_t)nTiles * sizeof(unsigned)));
                c->capPhase = (size_t)nTiles;
                c->phaseTiles = 0;
            }
            if (c->phaseTiles != nTiles) {       // first free-running frame of this geometry: every phase starts empty
                memset(c->phaseState, 0, sizeof c->phaseState);
                HIP_TRY(c, hipMemsetAsync(c->dPhaseCost, 0, (size_t)nTiles * 64 * sizeof(unsigned), c->phaseStream));
                c->phaseTiles = nTiles;
            }
            phase = ((p->frameCount % 64) + 64) % 64;
            if (c->phaseState[phase] == 1) {
                const hipError_t q = hipEventQuery(c->evPhase[phase]);
                (void)hipGetLastError();
                if (q == hipSuccess) c->phaseState[phase] = 2;
            }
            // (a phase whose sort has not been SEEN complete keeps the average order chosen above: no stream ever waits for a sort)
            if (c->phaseState[phase] == 2) sc.tileOrder = c->dPhaseOrder + (size_t)phase * nTiles;
            sc.tileCost = c->dPhaseCost + (size_t)phase * nTiles;
        }
    }
    if (timed) HIP_TRY(c, hipEventRecord(c->evStart, s));
    HIP_TRY(c, rt_launch_render(f, sc, dColor, dPos, dNormal, counter, c->variant, s, countMode));
    if (timed) {
        HIP_TRY(c, hipEventRecord(c->evStop, s));
        c->timed = true;
    }
    HIP_TRY(c, hipEventRecord(mine->last, s));      // every launch, on every stream: rt_set_scene orders behind it
    if (phase >= 0) {
        // this frame's costs -> its phase's order, behind every launch that may still read that order (this one included), on
        // the phase stream; the costs also join the all-phase average (dTileCost) that the periodic sort below works on
        for (int i = 0; i < c->nFbStreams; i++) HIP_TRY(c, hipStreamWaitEvent(c->phaseStream, c->fbStreams[i].last, 0));
        HIP_TRY(c, rt_launch_lpt_sort(c->dPhaseCost + (size_t)phase * nTiles, c->dPhaseSnap, c->dPhaseOrder + (size_t)phase * nTiles,
                                      nTiles, c->phaseStream, c->dTileCost));
        if (!c->evPhase[phase]) HIP_TRY(c, hipEventCreateWithFlags(&c->evPhase[phase], hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->evPhase[phase], c->phaseStream));
        c->phaseState[phase] = 1;
    }
    if (sortAfter) {
        // Re-sort on the first two frames of a geometry, then every fbPeriod-th.  Costs accumulate in between, so
        // the order follows each tile's AVERAGE cost over the period: with a free-running frameCount (which rotates
        // the bounce sample all pixels share) the last frame alone is a poor predictor of the next (C2, measured:
        // 0.63 ms/frame ordering by the last frame every 8th, 0.56 by the 8-frame average, 0.54 by the 32-frame one).
        if ((c->fbAge < 2 || (c->fbAge % c->fbPeriod) == 0) && !c->sortPending) {
            const int next = c->fbCur == 0 ? 1 : 0;
            // the buffer about to be rewritten was last read before the previous adoption: every stream's last
            // launch (this one included) is after those reads, and gives the sort this frame's costs
            for (int i = 0; i < c->nFbStreams; i++) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->fbStreams[i].last, 0));
            HIP_TRY(c, rt_launch_lpt_sort(c->dTileCost, c->dTileSnap, c->dTileOrder[next], nTiles, c->stream));
            if (!c->evSort[next]) HIP_TRY(c, hipEventCreateWithFlags(&c->evSort[next], hipEventDisableTiming));
            HIP_TRY(c, hipEventRecord(c->evSort[next], c->stream));
            c->fbNext = next;
            c->sortPending = true;
            c->sortAge = c->fbAge;
        }
        c->fbAge++;
        c->fbTiles = nTiles;
        c->fbTilesX = tilesX;
        c->fbBt = bt;
    }
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_create(rt_context **out, int deviceId) {
    if (!out) return RT_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || deviceId < 0 || deviceId >= n) return RT_ERR_NO_DEVICE;
    if (hipSetDevice(deviceId) != hipSuccess) return RT_ERR_NO_DEVICE;
    rt_context *c = new (std::nothrow) rt_context();
    if (!c) return RT_ERR_HIP;
    c->device = deviceId;
    if (const char *e = getenv("RT_FB_PERIOD")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 1024) c->fbPeriod = (unsigned)v;
    }
    if (const char *e = getenv("RT_DEBUG_PRED_CLASSES")) c->dbgWantCls = atoi(e) != 0;
    if (const char *e = getenv("RT_PRED_MIN_TILES")) c->predMinTiles = atoi(e);
    if (const char *e = getenv("RT_PHASE_ORDER")) c->phaseOn = atoi(e) != 0;
    if (const char *e = getenv("RT_ST_BUILD")) c->stOnePhase = strcmp(e, "full") == 0;
    if (const char *e = getenv("RT_ST_GEOM")) {        // "Kcube,Kplan,NB": measurement override of the shadow-table geometry
        int kc = 0, kp = 0, nb = 0;
        if (sscanf(e, "%d,%d,%d", &kc, &kp, &nb) == 3 && kc >= 4 && kc <= 128 && kp >= 4 && kp <= 256 && nb >= 1 && nb <= 128)
            c->stGeomSmall = c->stGeomLarge = RtShadowTabGeom{kc, kp, nb};
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->evStart) != hipSuccess || hipEventCreate(&c->evStop) != hipSuccess ||
        hipEventCreateWithFlags(&c->evScene, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void **)&c->dRayCounter, 32 * sizeof(unsigned long long)) != hipSuccess) {
        rt_destroy(c);
        return RT_ERR_HIP;
    }
    *out = c;
    return RT_OK;
}

int rt_destroy(rt_context *c) {
    if (!c) return RT_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->phaseStream) { (void)hipStreamSynchronize(c->phaseStream); (void)hipStreamDestroy(c->phaseStream); }
    for (int k = 0; k < 64; k++)
        if (c->evPhase[k]) (void)hipEventDestroy(c->evPhase[k]);
    for (void *b : {(void *)c->dPhaseCost, (void *)c->dPhaseOrder, (void *)c->dPhaseSnap})
        if (b) (void)hipFree(b);
    for (int i = 0; i < rt_context::kMaxStreams; i++) {
        if (c->fbStreams[i].last) {
            (void)hipEventSynchronize(c->fbStreams[i].last);
            (void)hipEventDestroy(c->fbStreams[i].last);
        }
        for (void *b : {(void *)c->fbStreams[i].predOrder, (void *)c->fbStreams[i].predCls, (void *)c->fbStreams[i].predCounts})
            if (b) (void)hipFree(b);
    }
    void *bufs[] = {c->dShadowTab, c->dObjects, c->dLights, c->dCompiled, c->dNoise, c->dSky, c->dColor, c->dPos, c->dNormal, c->dRayCounter,
                    c->dTileCost, c->dTileSnap, c->dTileOrder[0], c->dTileOrder[1], c->dBloom[0], c->dBloom[1], c->dSsaoDepth, c->dFrameAO[0], c->dFrameAO[1], c->dHistory[0], c->dHistory[1], c->dFrameDisplay};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (c->evStart) (void)hipEventDestroy(c->evStart);
    if (c->evStop) (void)hipEventDestroy(c->evStop);
    if (c->evScene) (void)hipEventDestroy(c->evScene);
    for (int k = 0; k < 2; k++)
        if (c->evSort[k]) (void)hipEventDestroy(c->evSort[k]);
    for (int k = 0; k < 2; k++) {
        if (c->evStage[k]) (void)hipEventDestroy(c->evStage[k]);
        if (c->hStage[k]) (void)hipHostFree(c->hStage[k]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RT_OK;
}

int rt_set_scene(rt_context *c, const void *objects, int nObj, const void *lights, int nLt) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (nObj < 0 || nLt < 0 || (nObj > 0 && !objects) || (nLt > 0 && !lights))
        return fail(c, RT_ERR_INVALID_ARG, "bad scene arguments");
    if (nObj > RT_MAX_OBJECTS || nLt > RT_MAX_LIGHTS) return fail(c, RT_ERR_TOO_LARGE, "scene exceeds RT_MAX_OBJECTS/RT_MAX_LIGHTS");
    HIP_TRY(c, hipSetDevice(c->device));
    // The reference re-specifies both SSBOs every frame whether or not anything moved (ImGUIManager.cpp:202, :338).  Identical
    // bytes leave the device scene, its compiled form and its shadow tables as they are (and the scheduler's view of "the same
    // frame as before" intact).
    {
        const size_t ob = (size_t)nObj * RT_OBJECT_STRIDE, lb = (size_t)nLt * RT_LIGHT_STRIDE;
        if (c->dCompiled && c->nObj == nObj && c->nLt == nLt && c->lastScene.size() == ob + lb &&
            (ob == 0 || memcmp(c->lastScene.data(), objects, ob) == 0) && (lb == 0 || memcmp(c->lastScene.data() + ob, lights, lb) == 0))
            return RT_OK;
        c->lastScene.clear();              // (refilled at the end: a failed update must not look like the current scene)
        c->sceneGen++;
    }
    int rc;
    if ((rc = ensure(c, &c->dObjects, &c->capObjects, (size_t)nObj * RT_OBJECT_STRIDE))) return rc;
    if ((rc = ensure(c, &c->dLights, &c->capLights, (size_t)nLt * RT_LIGHT_STRIDE))) return rc;
    if ((rc = ensure(c, &c->dCompiled, &c->capCompiledF4, rt_compiled_total_f4(nObj, nLt)))) return rc;
    // The scene buffers are about to be rewritten on the context's stream: order that behind the last launch of
    // EVERY stream frames have been issued on (one event per stream; a single shared event would only cover the
    // most recent one -- ADVICE r1).  The reference re-uploads its SSBOs every frame, so this is the common path.
    for (int i = 0; i < c->nFbStreams; i++)
        if (c->fbStreams[i].s != c->stream) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->fbStreams[i].last, 0));
    const size_t objBytes = (size_t)nObj * RT_OBJECT_STRIDE, ltBytes = (size_t)nLt * RT_LIGHT_STRIDE;
    const int k = (int)(c->stageSeq++ & 1u);
    if (c->stageUsed[k]) HIP_TRY(c, hipEventSynchronize(c->evStage[k]));
    if (objBytes + ltBytes > c->capStage[k]) {
        if (c->hStage[k]) HIP_TRY(c, hipHostFree(c->hStage[k]));
        c->hStage[k] = nullptr;
        c->capStage[k] = 0;
        size_t cap = objBytes + ltBytes < 65536 ? 65536 : objBytes + ltBytes;
        HIP_TRY(c, hipHostMalloc((void **)&c->hStage[k], cap, hipHostMallocDefault));
        c->capStage[k] = cap;
    }
    if (objBytes) memcpy(c->hStage[k], objects, objBytes);
    if (ltBytes) memcpy(c->hStage[k] + objBytes, lights, ltBytes);
    if (objBytes) HIP_TRY(c, hipMemcpyAsync(c->dObjects, c->hStage[k], objBytes, hipMemcpyHostToDevice, c->stream));
    if (ltBytes) HIP_TRY(c, hipMemcpyAsync(c->dLights, c->hStage[k] + objBytes, ltBytes, hipMemcpyHostToDevice, c->stream));
    if (!c->evStage[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->evStage[k], hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->evStage[k], c->stream));
    c->stageUsed[k] = true;
    HIP_TRY(c, rt_launch_compile_scene(c->dObjects, nObj, c->dLights, nLt, c->dCompiled, c->stream));
    // the lights' shadow tables of this scene (light-space candidate masks, rt_shadowtab.inc)
    c->shadowTabValid = false;
    if (nObj > 0 && nLt > 0 && nObj <= RT_ST_MAX_OBJECTS && nLt <= RT_ST_MAX_LIGHTS) {
        const RtShadowTabGeom &g = nObj <= 32 ? c->stGeomSmall : c->stGeomLarge;
        bool pcss = false;            // a PCSS light: its blocker rays get a table of their own (the PCSS kernel instantiations read it)
        for (int i = 0; i < nLt; i++) {
            rt_light l;
            memcpy(&l, (const uint8_t *)lights + (size_t)i * RT_LIGHT_STRIDE, sizeof l);
            pcss = pcss || l.shadowType == 2;
        }
        c->shadowTabBlocker = pcss;
        if ((rc = ensure(c, &c->dShadowTab, &c->capShadowTab, rt_shadowtab_dwords(g, nObj, nLt, pcss)))) return rc;
        HIP_TRY(c, rt_launch_shadow_tables(c->dCompiled, nObj, nLt, c->dShadowTab, g, c->stream, c->stOnePhase, pcss));
        c->shadowTabValid = true;
    }
    HIP_TRY(c, hipEventRecord(c->evScene, c->stream));   // foreign streams order behind this (rt_render_to)
    try {
        c->lastScene.resize(objBytes + ltBytes);
        if (objBytes) memcpy(c->lastScene.data(), objects, objBytes);
        if (ltBytes) memcpy(c->lastScene.data() + objBytes, lights, ltBytes);
    } catch (...) {
        c->lastScene.clear();
    }
    c->nObj = nObj;
    c->nLt = nLt;
    c->anyPcss = false;
    for (int i = 0; i < nLt; i++) {
        rt_light l;
        memcpy(&l, (const uint8_t *)lights + (size_t)i * RT_LIGHT_STRIDE, sizeof l);
        c->anyPcss = c->anyPcss || l.shadowType == 2;
    }
    return RT_OK;
}

int rt_set_noise(rt_context *c, const uint8_t *r8, int w, int h) {
    if (!c) return RT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, fb_sync_all(c));            // frames on every stream read the texture
    c->texGen++;
    if (c->dNoise) { HIP_TRY(c, hipFree(c->dNoise)); c->dNoise = nullptr; }
    c->noiseW = c->noiseH = 0;
    if (!r8) return RT_OK;
    if (w <= 0 || h <= 0) return fail(c, RT_ERR_INVALID_ARG, "noise size must be positive");
    HIP_TRY(c, hipMalloc((void **)&c->dNoise, (size_t)w * h));
    HIP_TRY(c, hipMemcpy(c->dNoise, r8, (size_t)w * h, hipMemcpyHostToDevice));
    c->noiseW = w;
    c->noiseH = h;
    return RT_OK;
}

int rt_set_skybox(rt_context *c, const uint16_t *rgb16f, int size) {
    if (!c) return RT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, fb_sync_all(c));
    c->texGen++;
    if (c->dSky) { HIP_TRY(c, hipFree(c->dSky)); c->dSky = nullptr; }
    c->skySize = 0;
    if (!rgb16f) return RT_OK;
    if (size <= 0) return fail(c, RT_ERR_INVALID_ARG, "skybox size must be positive");
    size_t bytes = (size_t)6 * size * size * 3 * sizeof(uint16_t);
    HIP_TRY(c, hipMalloc((void **)&c->dSky, bytes));
    HIP_TRY(c, hipMemcpy(c->dSky, rgb16f, bytes, hipMemcpyHostToDevice));
    c->skySize = size;
    return RT_OK;
}

int rt_render(rt_context *c, const rt_params *p) {
    if (!c) return RT_ERR_INVALID_ARG;
    int rc = validate_params(c, p);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    size_t npx = (size_t)p->regionW * p->regionH;
    if (npx > c->capPixels || !c->dColor) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->dColor) HIP_TRY(c, hipFree(c->dColor));
        if (c->dPos) HIP_TRY(c, hipFree(c->dPos));
        if (c->dNormal) HIP_TRY(c, hipFree(c->dNormal));
        c->dColor = c->dPos = nullptr;
        c->dNormal = nullptr;
        c->capPixels = 0;
        size_t n = npx ? npx : 1;
        HIP_TRY(c, hipMalloc((void **)&c->dColor, n * sizeof(float4)));
        HIP_TRY(c, hipMalloc((void **)&c->dPos, n * sizeof(float4)));
        HIP_TRY(c, hipMalloc((void **)&c->dNormal, n * sizeof(uint2)));
        c->capPixels = n;
    }
    c->surfW = p->regionW;
    c->surfH = p->regionH;
    return launch(c, p, c->dColor, c->dPos, c->dNormal, nullptr, c->stream, true);
}

int rt_render_to(rt_context *c, const rt_params *p, void *dColor, void *dPosition, void *dNormal, void *hipStream) {
    if (!c) return RT_ERR_INVALID_ARG;
    int rc = validate_params(c, p);
    if (rc) return rc;
    if (!dColor || !dPosition || !dNormal) return fail(c, RT_ERR_INVALID_ARG, "NULL device surface");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hipStream ? (hipStream_t)hipStream : c->stream;
    if (s != c->stream) HIP_TRY(c, hipStreamWaitEvent(s, c->evScene, 0));
    return launch(c, p, (float4 *)dColor, (float4 *)dPosition, (uint2 *)dNormal, nullptr, s, true);
}

int rt_render_into_image(rt_context *c, const rt_params *p, void *dColorImage, void *dPositionImage, void *dNormalImage, void *hipStream) {
    if (!c) return RT_ERR_INVALID_ARG;
    int rc = validate_params(c, p);
    if (rc) return rc;
    if (!dColorImage || !dPositionImage || !dNormalImage) return fail(c, RT_ERR_INVALID_ARG, "NULL device surface");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hipStream ? (hipStream_t)hipStream : c->stream;
    if (s != c->stream) HIP_TRY(c, hipStreamWaitEvent(s, c->evScene, 0));
    return launch(c, p, (float4 *)dColorImage, (float4 *)dPositionImage, (uint2 *)dNormalImage, nullptr, s, true, 1, true);
}

int rt_context_stream(rt_context *c, void **hipStream) {
    if (!c || !hipStream) return RT_ERR_INVALID_ARG;
    *hipStream = (void *)c->stream;
    return RT_OK;
}

int rt_sync(rt_context *c) {
    if (!c) return RT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_readback(rt_context *c, float *gColor, float *gPosition, uint16_t *gNormal) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!c->dColor || c->surfW <= 0 || c->surfH <= 0) return fail(c, RT_ERR_NO_SURFACES, "no rendered surfaces");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    size_t npx = (size_t)c->surfW * c->surfH;
    if (gColor) HIP_TRY(c, hipMemcpy(gColor, c->dColor, npx * sizeof(float4), hipMemcpyDeviceToHost));
    if (gPosition) HIP_TRY(c, hipMemcpy(gPosition, c->dPos, npx * sizeof(float4), hipMemcpyDeviceToHost));
    if (gNormal) HIP_TRY(c, hipMemcpy(gNormal, c->dNormal, npx * sizeof(uint2), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_get_surfaces(rt_context *c, void **dColor, void **dPosition, void **dNormal) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!c->dColor) return fail(c, RT_ERR_NO_SURFACES, "no rendered surfaces");
    if (dColor) *dColor = c->dColor;
    if (dPosition) *dPosition = c->dPos;
    if (dNormal) *dNormal = c->dNormal;
    return RT_OK;
}

int rt_last_kernel_ms(rt_context *c, float *ms) {
    if (!c || !ms) return RT_ERR_INVALID_ARG;
    if (!c->timed) return fail(c, RT_ERR_NO_SURFACES, "no timed render yet");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->evStop));
    HIP_TRY(c, hipEventElapsedTime(ms, c->evStart, c->evStop));
    return RT_OK;
}

static int count_rays_impl(rt_context *c, const rt_params *p, uint64_t *rays, int countMode);
int rt_count_rays(rt_context *c, const rt_params *p, uint64_t *rays) { return count_rays_impl(c, p, rays, 1); }
int rt_count_rays_traced(rt_context *c, const rt_params *p, uint64_t *rays) { return count_rays_impl(c, p, rays, 2); }
static int count_rays_impl(rt_context *c, const rt_params *p, uint64_t *rays, int countMode) {
    if (!c || !rays) return RT_ERR_INVALID_ARG;
    int rc = validate_params(c, p);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    size_t npx = (size_t)p->regionW * p->regionH;
    float4 *col = nullptr, *pos = nullptr;
    uint2 *nrm = nullptr;
    size_t n = npx ? npx : 1;
    HIP_TRY(c, hipMalloc((void **)&col, n * sizeof(float4)));
    hipError_t e1 = hipMalloc((void **)&pos, n * sizeof(float4));
    hipError_t e2 = hipMalloc((void **)&nrm, n * sizeof(uint2));
    if (e1 == hipSuccess && e2 == hipSuccess) {
        rc = RT_OK;
        if (hipMemsetAsync(c->dRayCounter, 0, 32 * sizeof(unsigned long long), c->stream) != hipSuccess) rc = RT_ERR_HIP;
        if (!rc) rc = launch(c, p, col, pos, nrm, c->dRayCounter, c->stream, false, countMode);
        unsigned long long v[32] = {};
        if (!rc && (hipStreamSynchronize(c->stream) != hipSuccess ||
                    hipMemcpy(v, c->dRayCounter, sizeof v, hipMemcpyDeviceToHost) != hipSuccess))
            rc = fail(c, RT_ERR_HIP, "ray counter readback");
        *rays = v[0];
        for (int k = 0; k < 32; k++) c->lastStats[k] = v[k];
    } else {
        rc = fail(c, RT_ERR_HIP, "hipMalloc (ray count scratch)", e1 != hipSuccess ? e1 : e2);
    }
    (void)hipFree(col);
    if (pos) (void)hipFree(pos);
    if (nrm) (void)hipFree(nrm);
    return rc;
}

int rt_set_variant(rt_context *c, int variant) {
    if (!c) return RT_ERR_INVALID_ARG;
    // bit 8 (0x100): raster tile order every frame; bit 9 (0x200): the measured-cost feedback order of rounds 1-2 instead of the
    // predicted one (default)
    c->variant = variant & 0xff;
    c->feedback = (variant & 0x100) == 0;
    c->schedMode = (variant & 0x100) ? 0 : ((variant & 0x200) ? 1 : 2);
    c->fbTiles = 0;
    return RT_OK;
}

int rt_debug_stats(rt_context *c, uint64_t out[4]) {
    if (!c || !out) return RT_ERR_INVALID_ARG;
    for (int k = 0; k < 4; k++) out[k] = c->lastStats[k];
    return RT_OK;
}

int rt_debug_stats_ex(rt_context *c, uint64_t out[32]) {
    if (!c || !out) return RT_ERR_INVALID_ARG;
    for (int k = 0; k < 32; k++) out[k] = c->lastStats[k];
    return RT_OK;
}

int rt_debug_shadow_tables(rt_context *c, uint32_t *out, size_t capDwords, size_t *nDwords, int *wordsPerCell) {
    if (!c || !nDwords) return RT_ERR_INVALID_ARG;
    *nDwords = 0;
    if (wordsPerCell) *wordsPerCell = 0;
    if (!c->shadowTabValid) return RT_OK;
    const RtShadowTabGeom &g = c->nObj <= 32 ? c->stGeomSmall : c->stGeomLarge;
    const size_t n = rt_shadowtab_table_dwords(g, c->nObj, c->nLt, c->shadowTabBlocker);      // (the builder's scratch behind the tables is not part of them)
    *nDwords = n;
    if (wordsPerCell) *wordsPerCell = rt_shadowtab_words(c->nObj);
    if (!out) return RT_OK;
    if (capDwords < n) return fail(c, RT_ERR_TOO_LARGE, "buffer smaller than the shadow tables");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, c->dShadowTab, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_debug_predicted_classes(rt_context *c, uint8_t *out, int cap, int *nTiles) {
    if (!c || !out || cap < 0 || !nTiles) return RT_ERR_INVALID_ARG;
    *nTiles = 0;
    if (!c->dbgPredCls) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, fb_sync_all(c));
    *nTiles = c->dbgPredTiles;
    const int n = *nTiles < cap ? *nTiles : cap;
    HIP_TRY(c, hipMemcpy(out, c->dbgPredCls, (size_t)n, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_debug_tile_costs(rt_context *c, unsigned *out, int cap, int *nTiles, int *tilesX) {
    if (!c || !out || cap < 0 || !nTiles || !tilesX) return RT_ERR_INVALID_ARG;
    *nTiles = c->fbTiles;
    *tilesX = c->fbTilesX;
    if (c->fbTiles == 0 || !c->dTileCost) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, fb_sync_all(c));
    const int n = c->fbTiles < cap ? c->fbTiles : cap;
    HIP_TRY(c, hipMemcpy(out, c->dTileCost, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_taa_resolve(rt_context *c, const void *dCurrent, const void *dHistory, const void *dNormal, void *dOut, int width,
                   int height, float blendFactor, float jitterX, float jitterY, void *hipStream) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!dCurrent || !dHistory || !dNormal || !dOut || width <= 0 || height <= 0)
        return fail(c, RT_ERR_INVALID_ARG, "bad rt_taa_resolve arguments");
    if (dOut == dCurrent || dOut == dHistory) return fail(c, RT_ERR_INVALID_ARG, "rt_taa_resolve cannot run in place");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hipStream ? (hipStream_t)hipStream : c->stream;
    HIP_TRY(c, rt_launch_taa_resolve(dCurrent, dHistory, dNormal, dOut, width, height, blendFactor, jitterX, jitterY, s));
    return RT_OK;
}

int rt_bloom(rt_context *c, const void *dScene, void *dOut, int width, int height, float threshold, float strength,
             int iterations, void *hipStream) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!dScene || !dOut || width <= 0 || height <= 0 || iterations < 0) return fail(c, RT_ERR_INVALID_ARG, "bad rt_bloom arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hipStream ? (hipStream_t)hipStream : c->stream;
    const size_t npx = (size_t)width * height;
    if (npx > c->capBloomPx) {
        HIP_TRY(c, hipStreamSynchronize(s));
        for (int k = 0; k < 2; k++) {
            if (c->dBloom[k]) HIP_TRY(c, hipFree(c->dBloom[k]));
            c->dBloom[k] = nullptr;
        }
        c->capBloomPx = 0;
        HIP_TRY(c, hipMalloc(&c->dBloom[0], npx * 8));
        HIP_TRY(c, hipMalloc(&c->dBloom[1], npx * 8));
        c->capBloomPx = npx;
    }
    HIP_TRY(c, rt_launch_bloom(dScene, c->dBloom[0], c->dBloom[1], dOut, width, height, threshold, strength, iterations, s));
    return RT_OK;
}

const char *rt_last_error(rt_context *c) { return c ? c->err.c_str() : "NULL context"; }

int rt_debug_mesa_math(const float *in, float *out, int n) {
    if (!in || !out || n < 0) return RT_ERR_INVALID_ARG;
    for (int i = 0; i < n; i++) {
        out[4 * i] = rtm::sin_(in[i]);
        out[4 * i + 1] = rtm::cos_(in[i]);
        out[4 * i + 2] = rtm::tan_(in[i]);
        out[4 * i + 3] = rtm::exp_(in[i]);
    }
    return RT_OK;
}

int rt_equirect_to_cubemap(rt_context *c, const float *hEquirectRGB, int width, int height, int size, void *dFacesOut,
                           int install) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!hEquirectRGB || width <= 0 || height <= 0 || size <= 0) return fail(c, RT_ERR_INVALID_ARG, "bad rt_equirect_to_cubemap arguments");
    if (!dFacesOut && !install) return fail(c, RT_ERR_INVALID_ARG, "nowhere to put the faces");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t npx = (size_t)width * height, faceBytes = (size_t)6 * size * size * 3 * sizeof(uint16_t);
    float *dRgb = nullptr;
    void *dTex = nullptr, *dFaces = nullptr;
    hipError_t e = hipMalloc((void **)&dRgb, npx * 3 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&dTex, npx * 8);
    if (e == hipSuccess) e = hipMalloc(&dFaces, faceBytes);
    if (e == hipSuccess) e = hipMemcpyAsync(dRgb, hEquirectRGB, npx * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = rt_launch_equirect_to_cubemap(dRgb, dTex, width, height, size, dFaces, c->stream);
    if (e == hipSuccess && dFacesOut) e = hipMemcpyAsync(dFacesOut, dFaces, faceBytes, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (dRgb) (void)hipFree(dRgb);
    if (dTex) (void)hipFree(dTex);
    if (e != hipSuccess) {
        if (dFaces) (void)hipFree(dFaces);
        return fail(c, RT_ERR_HIP, "rt_equirect_to_cubemap", e);
    }
    if (install) {
        c->texGen++;
        if (c->dSky) (void)hipFree(c->dSky);
        c->dSky = (decltype(c->dSky))dFaces;
        c->skySize = size;
    } else {
        (void)hipFree(dFaces);
    }
    return RT_OK;
}

int rt_ssao(rt_context *c, const void *dPosition, const void *dNormal, void *dOut, int width, int height, const float *hNoise,
            int noiseW, int noiseH, const float *hSamples, const float *hProjection, const float *hView, void *hipStream) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!dPosition || !dNormal || !dOut || !hNoise || !hSamples || !hProjection || !hView || width <= 0 || height <= 0)
        return fail(c, RT_ERR_INVALID_ARG, "bad rt_ssao arguments");
    if (noiseW <= 0 || noiseH <= 0 || noiseW * noiseH > 16) return fail(c, RT_ERR_TOO_LARGE, "rotation texture larger than 16 texels");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hipStream ? (hipStream_t)hipStream : c->stream;
    const size_t npx = (size_t)width * height;
    if (npx > c->capSsaoPx) {
        HIP_TRY(c, hipDeviceSynchronize());
        if (c->dSsaoDepth) HIP_TRY(c, hipFree(c->dSsaoDepth));
        c->dSsaoDepth = nullptr;
        c->capSsaoPx = 0;
        HIP_TRY(c, hipMalloc((void **)&c->dSsaoDepth, npx * sizeof(float)));
        c->capSsaoPx = npx;
    }
    HIP_TRY(c, rt_launch_ssao(dPosition, dNormal, c->dSsaoDepth, dOut, width, height, hNoise, noiseW, noiseH, hSamples, hProjection,
                              hView, s));
    return RT_OK;
}

int rt_ssao_blur(rt_context *c, const void *dIn, void *dOut, int width, int height, int horizontal, void *hipStream) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!dIn || !dOut || width <= 0 || height <= 0) return fail(c, RT_ERR_INVALID_ARG, "bad rt_ssao_blur arguments");
    if (dIn == dOut) return fail(c, RT_ERR_INVALID_ARG, "rt_ssao_blur cannot run in place");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hipStream ? (hipStream_t)hipStream : c->stream;
    HIP_TRY(c, rt_launch_ssao_blur(dIn, dOut, width, height, horizontal, s));
    return RT_OK;
}

int rt_frame(rt_context *c, const rt_params *p, const rt_frame_desc *d, void *dDisplay) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!d) return fail(c, RT_ERR_INVALID_ARG, "frame description is NULL");
    if (d->enableAO && (!d->aoSamples || !d->aoNoise)) return fail(c, RT_ERR_INVALID_ARG, "AO needs its kernel samples and rotation texture");
    if (d->bloomIterations < 0) return fail(c, RT_ERR_INVALID_ARG, "bloomIterations < 0");
    int rc = validate_params(c, p);
    if (rc) return rc;
    if (p->x0 != 0 || p->y0 != 0 || p->regionW != p->width || p->regionH != p->height || p->stripCycleRows != 0 || p->stripCount != 1)
        return fail(c, RT_ERR_INVALID_ARG, "rt_frame renders the whole image on one device");
    const int W = p->width, H = p->height;
    const size_t npx = (size_t)W * H;
    HIP_TRY(c, hipSetDevice(c->device));
    if (npx > c->capFramePx) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        void **bufs[] = {(void **)&c->dFrameAO[0], (void **)&c->dFrameAO[1], (void **)&c->dHistory[0], (void **)&c->dHistory[1],
                         (void **)&c->dFrameDisplay};
        for (void **b : bufs) {
            if (*b) HIP_TRY(c, hipFree(*b));
            *b = nullptr;
        }
        c->capFramePx = 0;
        HIP_TRY(c, hipMalloc((void **)&c->dFrameAO[0], npx * sizeof(float)));
        HIP_TRY(c, hipMalloc((void **)&c->dFrameAO[1], npx * sizeof(float)));
        HIP_TRY(c, hipMalloc((void **)&c->dHistory[0], npx * sizeof(float4)));
        HIP_TRY(c, hipMalloc((void **)&c->dHistory[1], npx * sizeof(float4)));
        HIP_TRY(c, hipMalloc((void **)&c->dFrameDisplay, npx * sizeof(float4)));
        c->capFramePx = npx;
        c->frameW = c->frameH = 0;
    }
    if (c->frameW != W || c->frameH != H) {      // new size: the history starts as zeros (glTexImage2D(..., nullptr))
        HIP_TRY(c, hipMemsetAsync(c->dHistory[0], 0, npx * sizeof(float4), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->dHistory[1], 0, npx * sizeof(float4), c->stream));
        c->frameW = W;
        c->frameH = H;
        c->lastHistory = -1;
    }
    rc = rt_render(c, p);                                                   // :155-182
    if (rc) return rc;
    c->frameAOValid = false;
    if (d->enableAO) {                                                      // :185-187, AO.cpp:86-117
        float view[16], proj[16];
        rc = rt_camera_matrices(p->camPos, p->camDir, p->camUp, p->fovDeg, (float)W / (float)H, view, proj);
        if (rc) return fail(c, rc, "rt_camera_matrices");
        rc = rt_ssao(c, c->dPos, c->dNormal, c->dFrameAO[0], W, H, d->aoNoise, 4, 4, d->aoSamples, proj, view, nullptr);
        if (rc) return rc;
        rc = rt_ssao_blur(c, c->dFrameAO[0], c->dFrameAO[1], W, H, 0, nullptr);   // `horizontal` is never set upstream
        if (rc) return rc;
        c->frameAOValid = true;
    }
    rc = rt_bloom(c, c->dColor, dDisplay ? dDisplay : (void *)c->dFrameDisplay, W, H, d->bloomThreshold, d->bloomStrength,
                  d->bloomIterations, nullptr);                             // :189-228
    if (rc) return rc;
    if (d->enableTAA) {                                                     // :231-258
        const int cur = ((p->frameCount % 2) + 2) % 2;
        float jx, jy;
        rc = rt_taa_jitter(p->frameCount, W, H, &jx, &jy);
        if (rc) return fail(c, rc, "rt_taa_jitter");
        rc = rt_taa_resolve(c, c->dColor, c->dHistory[1 - cur], c->dNormal, c->dHistory[cur], W, H, d->taaBlendFactor, jx, jy, nullptr);
        if (rc) return rc;
        c->lastHistory = cur;
    }
    return RT_OK;
}

int rt_frame_surfaces(rt_context *c, void **dColor, void **dPosition, void **dNormal, void **dAO, void **dHistory) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (dColor) *dColor = c->dColor;
    if (dPosition) *dPosition = c->dPos;
    if (dNormal) *dNormal = c->dNormal;
    if (dAO) *dAO = c->frameAOValid ? c->dFrameAO[1] : nullptr;
    if (dHistory) *dHistory = c->lastHistory >= 0 ? c->dHistory[c->lastHistory] : nullptr;
    return RT_OK;
}

int rt_strip_local_rows(int height, int stripRows, int stripCount, int stripIndex) {
    if (height < 0 || stripRows <= 0 || stripCount <= 0 || stripIndex < 0 || stripIndex >= stripCount) return RT_ERR_INVALID_ARG;
    int nStrips = (height + stripRows - 1) / stripRows;
    int rows = 0;
    for (int s = stripIndex; s < nStrips; s += stripCount) {
        int r0 = s * stripRows, r1 = r0 + stripRows;
        if (r1 > height) r1 = height;
        rows += r1 - r0;
    }
    return rows;
}

int rt_deinterleave(rt_context *c, const void *src, void *dst, int width, int height, int bytesPerPixel,
                    int stripRows, int stripCount, size_t rankStrideBytes, void *hipStream) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!src || !dst || width <= 0 || height <= 0 || bytesPerPixel <= 0 || stripRows <= 0 || stripCount <= 0)
        return fail(c, RT_ERR_INVALID_ARG, "bad deinterleave arguments");
    if (((size_t)width * bytesPerPixel) % 4 != 0 || rankStrideBytes % 4 != 0)
        return fail(c, RT_ERR_INVALID_ARG, "row bytes and rank stride must be multiples of 4");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hipStream ? (hipStream_t)hipStream : c->stream;
    HIP_TRY(c, rt_launch_deinterleave(src, dst, width, height, bytesPerPixel, stripRows, stripCount, rankStrideBytes, s));
    return RT_OK;
}

size_t rt_wire_bytes(size_t nPixels) { return (nPixels * 30 + 15) / 16 * 16; }

int rt_wire_pack(rt_context *c, const void *dColor, const void *dPosition, const void *dNormal, void *dWire, size_t nPixels,
                 void *hipStream) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!dColor || !dPosition || !dNormal || !dWire) return fail(c, RT_ERR_INVALID_ARG, "bad wire_pack arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hipStream ? (hipStream_t)hipStream : c->stream;
    HIP_TRY(c, rt_launch_wire_pack(dColor, dPosition, dNormal, dWire, nPixels, s));
    return RT_OK;
}

int rt_wire_unpack(rt_context *c, const void *dWire, size_t rankStrideBytes, size_t rankPixels, const void *dRootColor,
                   const void *dRootPosition, const void *dRootNormal, int rootStrips, void *dColor, void *dPosition,
                   void *dNormal, int width, int height, int stripRows, int stripCount, void *hipStream) {
    if (!c) return RT_ERR_INVALID_ARG;
    if (!dWire || !dColor || !dPosition || !dNormal || width <= 0 || height <= 0 || stripRows <= 0 || stripCount <= 0 ||
        rootStrips <= 0)
        return fail(c, RT_ERR_INVALID_ARG, "bad wire_unpack arguments");
    const bool rootLocal = dRootColor || dRootPosition || dRootNormal;
    if (rootLocal && !(dRootColor && dRootPosition && dRootNormal))
        return fail(c, RT_ERR_INVALID_ARG, "the root's three local surfaces must be given together");
    if (!rootLocal && rootStrips != 1) return fail(c, RT_ERR_INVALID_ARG, "a larger root share needs the root's local surfaces");
    // every rank buffer must hold the rank's whole strips, and the f32 / f16 planes must stay aligned
    const int cycleRows = (rootStrips + stripCount - 1) * stripRows;
    const int nCycles = (height + cycleRows - 1) / cycleRows;
    const size_t needPixels = (size_t)nCycles * stripRows * width;
    if (rankPixels < needPixels || rankStrideBytes % 4 != 0 || rankStrideBytes < rankPixels * 30)
        return fail(c, RT_ERR_INVALID_ARG, "rank buffers too small or misaligned for this strip plan");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hipStream ? (hipStream_t)hipStream : c->stream;
    HIP_TRY(c, rt_launch_wire_unpack(dWire, rankStrideBytes, rankPixels, dRootColor, dRootPosition, dRootNormal,
                                     rootStrips * stripRows, dColor, dPosition, dNormal, width, height, stripRows, stripCount, s));
    return RT_OK;
}

}  // extern "C"
