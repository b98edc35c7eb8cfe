/*
 * rt_mi355.h -- C ABI of librt_mi355.so: the MI355X (gfx950) replacement for the
 * reference's ray-tracing compute dispatch.
 *
 * The reference has no FFI; its seam is the GL call sequence in
 * ForwardShadingPipline::Render() (/root/reference/src/ForwardShadingPipeline.cpp:155-182):
 * upload two SSBO byte arrays, set the uniforms, bind the cubemap, glDispatchCompute,
 * glMemoryBarrier -- after which three textures hold the result.  Every entry point
 * below names the reference call(s) it stands in for.  Plain pointers and sizes only;
 * no torch / STL types cross this boundary.  All functions return 0 on success or a
 * negative rt_status; nothing throws or aborts across the ABI.
 *
 * Threading: a context is owned by one thread at a time (the reference likewise has
 * one GL context on one thread).  Multi-GPU = one context per device / process.
 */
#ifndef RT_MI355_H
#define RT_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OBJECT_STRIDE 176 /* sizeof(Object), /root/reference/src/Object.h:13-21 */
#define RT_LIGHT_STRIDE 96   /* sizeof(Light),  /root/reference/src/Light.h:7-20  */
/* The shader's SSBOs are runtime-sized arrays whose lengths come from uniforms (raytracingCs.glsl:65-73); the default kernel
 * takes any count up to these sanity caps.  The exhaustive cross-check kernel (rt_set_variant 0) stages the whole scene in LDS
 * and refuses scenes beyond RT_EXHAUSTIVE_MAX_* with RT_ERR_TOO_LARGE at render time. */
#define RT_MAX_OBJECTS 65536
#define RT_MAX_LIGHTS 4096
#define RT_EXHAUSTIVE_MAX_OBJECTS 512   /* 512 * 160 B = 80 KiB of the CU's 160 KiB LDS */
#define RT_EXHAUSTIVE_MAX_LIGHTS 64

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = -1,
    RT_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime error on create */
    RT_ERR_HIP = -3,         /* a HIP call failed; see rt_last_error */
    RT_ERR_TOO_LARGE = -4,   /* scene exceeds RT_MAX_OBJECTS / RT_MAX_LIGHTS (or RT_EXHAUSTIVE_MAX_* under variant 0) */
    RT_ERR_NO_SURFACES = -5, /* readback before any render */
    RT_ERR_PARSE = -6
} rt_status;

/* Host mirror of the std430 records (offsets asserted below).
 * The ABI takes raw bytes; these structs exist so C/C++ callers can fill them. */
typedef struct rt_material { /* /root/reference/src/Material.h:11-23 */
    int32_t type;            /* never read by the shader */
    int32_t _pad0[3];
    float albedo[3];
    float metallic;
    float roughness;
    float diffuseStrength;
    float ior;
    float transparency;
    float specular;          /* never read by the shader */
    float subsurfaceScatter;
    int32_t _pad1[2];
    float subsurfaceColor[3];
    float scatterDistance;
} rt_material;

typedef struct rt_object { /* /root/reference/src/Object.h:13-21 */
    int32_t type;          /* 0 sphere, 1 plane */
    int32_t _pad0[3];
    float position[3];
    float radius;
    float normal[3];
    int32_t _pad1;
    float size[2];
    int32_t _pad2[2];
    rt_material material;
    float boundsMin[3];
    int32_t _pad3;
    float boundsMax[3];
    int32_t _pad4;
} rt_object;

typedef struct rt_light { /* /root/reference/src/Light.h:7-20 */
    int32_t type;         /* 0 point, 1 directional, 2 area */
    int32_t _pad0[3];
    float position[3];
    int32_t _pad1;
    float direction[3];
    int32_t _pad2;
    float color[3];
    float intensity;
    float radius;         /* dead in the shader */
    int32_t samples;      /* dead in the shader */
    float shadowSoftness;
    int32_t shadowType;   /* 0 none, 1 PCF, 2 PCSS */
    int32_t pcfSamples;
    float lightSize;
    float angularRadius;  /* dead in the shader */
    int32_t _pad3;
} rt_light;

/* The shader's uniform block (raytracingCs.glsl:72-89; uploaded by name at
 * ForwardShadingPipeline.cpp:155-166) + imageSize(outputImage) (:200) + the render
 * window.  Layout is shared with oracle/rt_oracle.h's orc_params. */
typedef struct rt_params {
    float camPos[3], camDir[3], camUp[3], camRight[3];
    float fovDeg;         /* degrees; Camera::FOV */
    float focalLength;    /* shader default 1.0 (never uploaded by the reference) */
    float maxRayDistance; /* shader default 114514.0 (never uploaded) */
    float noiseScale[2];  /* reference uploads 1/1024 */
    int32_t frameCount;
    int32_t useSkybox;
    int32_t maxRayDepth;  /* #define MAX_RAY_DEPTH (3 as shipped) */
    int32_t width, height;/* full image size: uv depends on it even for a window */
    /* Window rendered by this call, in local-row space.  The output surfaces are
     * regionW x regionH, row-major, row 0 = bottom (GL origin). */
    int32_t x0, y0, regionW, regionH;
    /* Interleaved row strips for multi-GPU tiling: local row ly is image row
     * ((ly / stripRows) * stripCount + stripIndex) * stripRows + ly % stripRows.
     * {1,1,0} is the identity (single GPU). */
    int32_t stripRows, stripCount, stripIndex;
    int32_t reserved0;
    /* Unequal strips (a rank that owns stripRows rows out of every stripCycleRows, starting at
     * stripOffsetRows): when stripCycleRows > 0 local row ly is image row
     * (ly / stripRows) * stripCycleRows + stripOffsetRows + ly % stripRows
     * and stripCount / stripIndex are ignored.  0 = the equal-strip rule above, which is the same
     * formula with stripCycleRows = stripRows * stripCount, stripOffsetRows = stripIndex * stripRows. */
    int32_t stripCycleRows, stripOffsetRows;
} rt_params;

typedef struct rt_context rt_context;

/* ---- lifetime: ForwardShadingPipline::Init() / dtor (ForwardShadingPipeline.cpp:6-20,
 *      .h:38-48): shader "compile" + stream + timing events on HIP device deviceId. */
int rt_create(rt_context **out, int deviceId);
int rt_destroy(rt_context *ctx);

/* ---- SSBO::update() + LightSSBO::update() (/root/reference/src/SSBO.h:16-23,
 *      LightSSBO.h:16-25): copies nObj*176 + nLt*96 bytes; caller keeps ownership; cheap
 *      enough to call every frame as the reference does (ImGUIManager.cpp:202,338). */
int rt_set_scene(rt_context *ctx, const void *objects, int nObj, const void *lights, int nLt);

/* ---- InitBlueNoiseTex (ForwardShadingPipeline.cpp:39-49): R8 texels, NEAREST/REPEAT.
 *      NULL = the shipped behaviour (sampler reads 0). */
int rt_set_noise(rt_context *ctx, const uint8_t *r8, int w, int h);

/* ---- glBindTexture(GL_TEXTURE_CUBE_MAP, ...) (ForwardShadingPipeline.cpp:167-170):
 *      6 faces (+X,-X,+Y,-Y,+Z,-Z) of size^2 RGB fp16, the format
 *      TextureLoader.cpp:140-147 allocates.  NULL unbinds. */
int rt_set_skybox(rt_context *ctx, const uint16_t *rgb16f, int size);

/* ---- glDispatchCompute + glMemoryBarrier (ForwardShadingPipeline.cpp:175-180).
 *      Asynchronous on the context's stream; renders into context-owned surfaces
 *      (outputImage rgba32f, gPosition rgba32f, gNormal rgba16f; raytracingCs.glsl:61-63). */
int rt_render(rt_context *ctx, const rt_params *p);

/* Same, into caller-provided DEVICE surfaces (regionW*regionH*16, *16, *8 bytes) on a
 * caller-provided hipStream_t (NULL = the context's stream).  Used for interop and for
 * the multi-GPU strip buffers that RCCL gathers. */
int rt_render_to(rt_context *ctx, const rt_params *p, void *dColor, void *dPosition,
                 void *dNormal, void *hipStream);

/* Same dispatch, but the three surfaces are WHOLE width x height images and every pixel of the rendered window (p->x0.., the
 * strip mapping) is stored at its image position (row-major, row 0 = bottom).  Several contexts -- one per GPU of a node, each
 * rendering its interleaved strips -- can thus fill ONE frame in place: on peer-mapped devices the stores of the other GPUs go
 * straight over xGMI into the frame's memory (rt_mgpu_render below does exactly that).  Pixels outside the image are skipped. */
int rt_render_into_image(rt_context *ctx, const rt_params *p, void *dColorImage, void *dPositionImage, void *dNormalImage,
                         void *hipStream);
/* The context's own hipStream_t (the stream rt_render / rt_set_scene use). */
int rt_context_stream(rt_context *ctx, void **hipStream);

/* ---- glFinish (PerformanceProfiler.cpp:51). */
int rt_sync(rt_context *ctx);

/* ---- glGetTexImage equivalents: host copies of the last rt_render's surfaces.
 *      Any pointer may be NULL. gNormal is 4 x fp16 per pixel (round-toward-zero). */
int rt_readback(rt_context *ctx, float *gColor, float *gPosition, uint16_t *gNormal);

/* Device pointers of the context-owned surfaces of the last rt_render. */
int rt_get_surfaces(rt_context *ctx, void **dColor, void **dPosition, void **dNormal);

/* ---- PerformanceProfiler Begin/EndGPUSection(RayTracing)
 *      (ForwardShadingPipeline.cpp:172-182, PerformanceProfiler.cpp:23-31): duration of
 *      the last render kernel from hipEvents on the launch stream; synchronises. */
int rt_last_kernel_ms(rt_context *ctx, float *ms);

/* Exact number of intersectObjects calls ("rays") for this frame,
 * counted by an instrumented build of the same kernel.  Synchronises. */
int rt_count_rays(rt_context *ctx, const rt_params *p, uint64_t *rays);
/* Rays the PRODUCTION kernel actually traverses for this frame: the same instrumented build, but keeping the
 * skips the timed kernel applies to rays whose result provably cannot reach a pixel (shadow / PCSS-blocker rays of
 * lanes whose light term is +-0 or NaN for every finite shadow value, blocker rays after the first blocker; DESIGN.md
 * section 4 items 6-8).  <= rt_count_rays; equal for variant 0.  Synchronises. */
int rt_count_rays_traced(rt_context *ctx, const rt_params *p, uint64_t *rays);

/* Kernel variant: 1 (default) = wavefront-packet kernel (packet culling, scalar-fed traversal),
 * 0 = exhaustive per-lane loop over all objects.  Both produce bit-identical surfaces; the switch
 * exists for A/B measurements and as a cross-check in the tests.  Adding 0x100
 * disables the cost-feedback tile order of the packet kernel (frame k's measured tile costs give
 * frame k+1's longest-first workgroup order; scheduling only, no pixel depends on it). */
int rt_set_variant(rt_context *ctx, int variant);

/* Diagnostics of the last rt_count_rays launch: out[0] rays, out[1] wave-level ray packets,
 * out[2] candidate objects summed over packets (after packet culling), out[3] 64-object cull
 * passes.  [1..3] are zero for variant 0 (no packet culling). */
int rt_debug_stats(rt_context *ctx, uint64_t out[4]);
/* The same plus packet-coherence diagnostics of the packet kernel: out[4..7] packets whose direction boxes leave 3 / 2 / 1 / 0
 * axes usable for culling (an axis is lost when the packet's directions straddle zero on it), out[8] packets that cannot be
 * culled at all (NaN lanes, non-finite origins), out[9] / out[10] candidates summed over the 3-axis packets / the others,
 * out[11] active lanes summed over packets; out[12..19] section timers of a -DRT_PK_TIMERS=1 build (shader clocks summed over
 * waves; zero otherwise): closest-hit traversals, a light's packet + candidate masks, its PCF sample loops, the whole wave,
 * the light packet's set-up alone, masks of octant-split light packets, split / unsplit light packets; out[20] / out[21]
 * candidates entering / leaving the per-lane cull level of a light's PCF rays; timers build only: out[22] candidate trips of
 * the PCF sample loops, out[23] those in which some lane passes the slab test, out[24] lanes passing, out[25] sample groups, out[26] the whole
 * light loop, out[27] the subsurface section, out[28] PCSS blocker searches, out[29] hit shading set-up, out[30] roulette + next
 * direction (clocks). */
int rt_debug_stats_ex(rt_context *ctx, uint64_t out[32]);
/* Measured cost (shader clock cycles / 64, summed over the tile's waves) of every workgroup tile of the
 * last feedback-scheduled rt_render / rt_render_to launch, in raster tile order; synchronises.  Writes up
 * to cap entries, *nTiles / *tilesX describe the tile grid.  Measurement hook, no reference counterpart. */
int rt_debug_tile_costs(rt_context *ctx, unsigned *out, int cap, int *nTiles, int *tilesX);

/* Cost classes (0 lightest .. 31 heaviest, ratio 2^(1/4)) the tile-order predictor gave the tiles of the last predicted
 * launch, raster tile order; *nTiles = capacity of the class buffer (>= that launch's tiles).  Measurement hook; synchronises. */
int rt_debug_predicted_classes(rt_context *ctx, uint8_t *out, int cap, int *nTiles);

/* The lights' SHADOW TABLES of the current scene (built by rt_set_scene for scenes of <= 256 objects; csrc/rt_shadowtab.inc):
 * per light 28 dwords of header -- (kind, base dword, K, NB) (invBinW, binMax, wlo, nmax^2) (pmin, qmin, invCell, cells)
 * (T0, eta) (B0, nmax) (light position or direction, binW) (base dword of the light's BLOCKER table or 0, its eta, -, -);
 * kind 0 no table, 1 cube map x distance bins (point / area), 2 planar grid x depth bins (directional) -- followed by the
 * cells, *wordsPerCell dwords each, bit i = object i may occlude a PCF ray of a shading point that reads the cell; scenes
 * with a PCSS light: a second table per such light, same cells, for pcssShadow's 16 blocker rays
 * (raytracingCs.glsl:409-427).  out == NULL only queries *nDwords.  Test hook
 * (tests/test_shadow_tables.py checks the tables against brute-force rays); synchronises.  No reference counterpart. */
int rt_debug_shadow_tables(rt_context *ctx, uint32_t *out, size_t capDwords, size_t *nDwords, int *wordsPerCell);

/* The GL driver's own sin / cos / tan / exp as the path evaluates them (csrc/rt_mesa_math.h: tan(radians(fov)/2) of
 * raytracingCs.glsl:209, cos/sin of :296-298, sin of random() :274, exp of :334), host side: out[4i..4i+3] =
 * sin, cos, tan, exp of in[i].  Test hook (no GPU needed), no reference counterpart. */
int rt_debug_mesa_math(const float *in, float *out, int n);

const char *rt_last_error(rt_context *ctx);

/* ---- host-side feeders of the byte contract (no GPU needed) */
/* GenerateAABBForObject (/root/reference/src/SceneIO.h:75-104), in place on n records. */
int rt_generate_aabb(void *objects, int n);
/* Camera::UpdateVectors (/root/reference/src/Camera.h:26-34). */
int rt_camera_vectors(float yawDeg, float pitchDeg, float front[3], float right[3], float up[3]);
/* SceneIO::Load's parser (/root/reference/src/SceneIO.h:108-122,145-186) on an in-memory
 * text; fills up to maxObj/maxLt records (AABBs generated), returns counts. */
int rt_scene_parse(const char *text, void *objects, int maxObj, int *nObj, void *lights,
                   int maxLt, int *nLt);

/* SceneIO::Save's writer (/root/reference/src/SceneIO.h:50-73, 124-142) into a caller buffer:
 * one "OBJECT <TYPE> <name> 18 numbers" / "LIGHT <TYPE> <name> 12 numbers" line per record, default
 * ostream float formatting.  Names may be NULL ("Object<i>" / "Light<i>").  *needed receives the
 * size including the terminator; out == NULL only queries it; too small -> RT_ERR_TOO_LARGE.
 * Lossy exactly like the reference's format (diffuseStrength, subsurface*, shadow fields are not stored). */
int rt_scene_write(const void *objects, int nObj, const void *lights, int nLt,
                   const char *const *objNames, const char *const *lightNames, char *out,
                   size_t cap, size_t *needed);

/* ---- next row after the ray tracer: the TAA resolve pass
 *      (/root/reference/shader/taaFs.glsl:13-53; host side ForwardShadingPipeline.cpp:231-260).
 *      dCurrent = gColor of this frame (rgba32f, sampled LINEAR/REPEAT), dHistory = the previous
 *      resolve (rgba32f, LINEAR/CLAMP_TO_EDGE), dNormal = gNormal (rgba16f, NEAREST/REPEAT), all
 *      width x height device surfaces as produced by rt_render; dOut = the new history (rgba32f).
 *      The caller ping-pongs dHistory/dOut like historyTex[2] (:233, :248).  Asynchronous on
 *      hipStream (NULL = the context's stream). */
int rt_taa_resolve(rt_context *ctx, const void *dCurrent, const void *dHistory, const void *dNormal,
                   void *dOut, int width, int height, float blendFactor, float jitterX,
                   float jitterY, void *hipStream);
/* uJitterX / uJitterY of ForwardShadingPipeline.cpp:241-242 (global.cpp:41-51's haltonSequence). */
int rt_taa_jitter(int frameCount, int width, int height, float *jitterX, float *jitterY);

/* ---- bloom: brightness extract (threshold), `iterations` alternating 9-tap
 *      Gaussian passes on rgba16f targets starting horizontal, combine scene + bloom*strength
 *      (/root/reference/shader/{brightness_extractFS,gaussian_blurFs,bloom_combineFs}.glsl;
 *      ForwardShadingPipeline.cpp:189-228 uses threshold 1.0, 10 iterations, strength 0.5).
 *      dScene = gColor (rgba32f), dOut = combined rgba32f (what the reference draws to the default
 *      framebuffer, before display quantisation); may not alias.  Asynchronous on hipStream. */
int rt_bloom(rt_context *ctx, const void *dScene, void *dOut, int width, int height, float threshold,
             float strength, int iterations, void *hipStream);

/* ---- next row: SSAO -- shader/ssaoFs.glsl:16-46 and ssao_blurFs.glsl:11-29 as driven by AOManager::RenderSSAO
 *      (AO.cpp:86-117).  dPosition (rgba32f) / dNormal (rgba16f) are the ray kernel's G-buffer surfaces;
 *      hNoise = the rotation texture (nW x nH rgba32f texels, nW*nH <= 16; AO.cpp:38-51 makes it 4x4),
 *      hSamples = the 64 kernel samples (AO.cpp:23-36), hProjection / hView = column-major mat4 (the glm
 *      matrices AO.cpp:91-92 uploads; rt_camera_matrices builds them like Camera.h:36-42); all four are HOST
 *      pointers, copied at the call.  dOut = width*height floats: the value the fragment shader writes (the
 *      reference renders it into FBOs without attachments, so upstream nothing consumes it).
 *      rt_ssao_blur: one separable 9-tap pass (the reference draws a single pass and never sets `horizontal`,
 *      i.e. vertical).  Asynchronous on hipStream. */
int rt_ssao(rt_context *ctx, const void *dPosition, const void *dNormal, void *dOut, int width, int height,
            const float *hNoise, int noiseW, int noiseH, const float *hSamples, const float *hProjection,
            const float *hView, void *hipStream);
int rt_ssao_blur(rt_context *ctx, const void *dIn, void *dOut, int width, int height, int horizontal, void *hipStream);
/* glm::lookAt(Position, Position + Front, Up) and glm::perspective(radians(FOV), aspect, 0.1, 100) of
 * Camera.h:36-42, column-major, fp32. */
int rt_camera_matrices(const float position[3], const float front[3], const float up[3], float fovDeg, float aspect,
                       float view[16], float projection[16]);

/* ---- next row: equirectangular -> cubemap -- ConvertHDRToCubemap (TextureLoader.cpp:118-194) with
 *      shader/skyboxVs.glsl + skyboxFs.glsl: hEquirectRGB = width*height*3 HOST floats as stbi_loadf returns
 *      them after the vertical flip (row 0 = bottom); the map is stored as RGB16F (rounded toward zero, as the
 *      reference's GL does on upload) and
 *      sampled LINEAR / CLAMP_TO_EDGE into six size x size RGB16F faces (GL face order, the layout
 *      rt_set_skybox takes).  dFacesOut (device, 6*size*size*3 halfs) may be NULL; install != 0 makes the
 *      result the context's skybox (what LoadHDRAsCubemap's caller does with the GL texture).  Synchronous. */
int rt_equirect_to_cubemap(rt_context *ctx, const float *hEquirectRGB, int width, int height, int size,
                           void *dFacesOut, int install);

/* ---- the caller of the path: one iteration of ForwardShadingPipline::Render()'s GPU work
 *      (ForwardShadingPipeline.cpp:155-260) in one call, on the context's own surfaces and stream:
 *        ray trace (:155-182)  ->  AO (:185-187, if enableAO)  ->  bloom: extract, blur passes, combine into the
 *        image the reference draws to the default framebuffer (:189-228)  ->  TAA resolve into
 *        historyTex[frameCount % 2] from historyTex[1 - frameCount % 2] (:231-258, if enableTAA).
 *      p->frameCount plays the reference's static frameCount (:141-142): it selects the history slot and the
 *      TAA jitter and is the shader's frameCount uniform; like the reference, the caller advances it only on
 *      frames with TAA enabled (:254).  History surfaces start as zeros.  dDisplay (device, width*height
 *      rgba32f) receives the bloom-combined image; may be NULL.  Asynchronous on the context's stream;
 *      rt_frame_surfaces returns the context-owned results (valid until the next size change). */
typedef struct rt_frame_desc {
    int32_t enableAO, enableTAA;
    float taaBlendFactor;            /* imguiManager.GetTAABlendFactor() */
    float bloomThreshold;            /* 1.0  (:197) */
    float bloomStrength;             /* 0.5  (:223) */
    int32_t bloomIterations;         /* 10   (:212) */
    const float *aoSamples;          /* 64 x 3 host floats (AO.cpp:23-36); required when enableAO */
    const float *aoNoise;            /* 4 x 4 x 4 host floats (AO.cpp:38-51) */
    float camYawPitchUnused[2];      /* reserved, zero */
} rt_frame_desc;
int rt_frame(rt_context *ctx, const rt_params *p, const rt_frame_desc *desc, void *dDisplay);
/* dAO = blurred AO (floats), dHistory = the history slot the last rt_frame wrote (rgba32f); NULL when that
 * pass has not run. */
int rt_frame_surfaces(rt_context *ctx, void **dColor, void **dPosition, void **dNormal, void **dAO, void **dHistory);

/* ---- multi-GPU strip helpers */
/* Number of local rows a rank owns for interleaved strips. */
int rt_strip_local_rows(int height, int stripRows, int stripCount, int stripIndex);
/* Rank-0 reassembly of gathered strip buffers.  src holds stripCount per-rank buffers,
 * rankStrideBytes apart, each made of whole strips of `width` pixels of bytesPerPixel (rows in
 * the rank's local order); dst = the full width x height image.  Device pointers;
 * asynchronous on hipStream (NULL = the context's stream). */
int rt_deinterleave(rt_context *ctx, const void *src, void *dst, int width, int height,
                    int bytesPerPixel, int stripRows, int stripCount, size_t rankStrideBytes,
                    void *hipStream);

/* Wire format for the gather (30 bytes per pixel instead of 40): every surface's alpha is the
 * constant 1.0 (raytracingCs.glsl:581-583), so a rank ships only
 *     [ gColor rgb f32 x nPixels | gPosition rgb f32 x nPixels | gNormal rgb f16 x nPixels ]
 * (rt_wire_bytes(nPixels) bytes, padded to 16) and rank 0 restores rgba with alpha = 1.0 while it
 * puts the strips back in image order.  No counterpart in the reference (single GPU).
 * rt_wire_pack: this rank's three surfaces (nPixels each, any row order) -> dWire.
 * rt_wire_unpack: dWire = stripCount rank buffers rankStrideBytes apart, each packed from
 * rankPixels pixels (whole strips of `width`); dst* = full width x height surfaces.  The image is
 * made of cycles of rootStrips strips of rank 0 followed by one strip of each other rank
 * (rootStrips = 1: the equal interleave).  With dRootColor/dRootPosition/dRootNormal != NULL rank
 * 0's rows are copied from its own local rgba surfaces (they never travel, wire slot 0 is ignored
 * and rank 0 may own a larger share: its rt_params use stripRows = rootStrips * stripRows,
 * stripCycleRows = (rootStrips + stripCount - 1) * stripRows, stripOffsetRows = 0); with NULL they
 * come from wire slot 0 and rootStrips must be 1.
 * Device pointers; asynchronous on hipStream (NULL = the context's stream). */
size_t rt_wire_bytes(size_t nPixels);
int rt_wire_pack(rt_context *ctx, const void *dColor, const void *dPosition, const void *dNormal, void *dWire,
                 size_t nPixels, void *hipStream);
int rt_wire_unpack(rt_context *ctx, const void *dWire, size_t rankStrideBytes, size_t rankPixels,
                   const void *dRootColor, const void *dRootPosition, const void *dRootNormal, int rootStrips,
                   void *dColor, void *dPosition, void *dNormal, int width, int height, int stripRows, int stripCount,
                   void *hipStream);

/* ---- one frame on N GPUs of a node from ONE process and ONE host thread -- what the reference's host is
 *      (/root/reference/src/main.cpp:3-7, ForwardShadingPipeline.cpp:129-271).  One context per entry of deviceIds (entries
 *      may repeat: N "devices" that are all device 0 run the N-way plan on a one-GPU box); device deviceIds[0] owns the frame.
 *      rt_mgpu_render splits the frame into interleaved strips of stripRows rows (default 8, rt_mgpu_set_strip_rows), every
 *      device renders its strips and its kernel stores them straight into the root's full-frame surfaces over xGMI (peer
 *      access; csrc/rt_mgpu.cpp) -- no gather buffer, no collective.  Asynchronous: the root context's stream (returned by
 *      rt_mgpu_get_surfaces) is ordered behind every device's stores, so work enqueued on it afterwards sees the whole frame;
 *      rt_mgpu_sync / rt_mgpu_readback wait for it.  p describes the WHOLE frame (identity window and strip fields).
 *      rt_mgpu_last_ms: duration of every device's share of the last frame (HIP events on its stream). */
typedef struct rt_mgpu rt_mgpu;
int rt_mgpu_create(rt_mgpu **out, const int *deviceIds, int nDevices);
int rt_mgpu_destroy(rt_mgpu *m);
int rt_mgpu_device_count(rt_mgpu *m);
int rt_mgpu_set_scene(rt_mgpu *m, const void *objects, int nObj, const void *lights, int nLt);
int rt_mgpu_set_noise(rt_mgpu *m, const uint8_t *r8, int w, int h);
int rt_mgpu_set_skybox(rt_mgpu *m, const uint16_t *rgb16f, int size);
int rt_mgpu_set_strip_rows(rt_mgpu *m, int stripRows);
int rt_mgpu_render(rt_mgpu *m, const rt_params *p);
int rt_mgpu_sync(rt_mgpu *m);
int rt_mgpu_get_surfaces(rt_mgpu *m, void **dColor, void **dPosition, void **dNormal, void **rootStream);
int rt_mgpu_readback(rt_mgpu *m, float *gColor, float *gPosition, uint16_t *gNormal);
int rt_mgpu_last_ms(rt_mgpu *m, float *perDeviceMs, int cap);
const char *rt_mgpu_last_error(rt_mgpu *m);

#ifdef __cplusplus
}
#endif

#if defined(__cplusplus) || (defined(__STDC_VERSION__) && __STDC_VERSION__ >= 201112L)
#ifdef __cplusplus
#define RT_SA(c, m) static_assert(c, m)
#else
#define RT_SA(c, m) _Static_assert(c, m)
#endif
RT_SA(sizeof(rt_material) == 80, "Material is 80 B");
RT_SA(sizeof(rt_frame_desc) == 32 + 2 * sizeof(void *), "rt_frame_desc layout");
RT_SA(sizeof(rt_object) == RT_OBJECT_STRIDE, "Object stride is 176 B");
RT_SA(sizeof(rt_light) == RT_LIGHT_STRIDE, "Light stride is 96 B");
RT_SA(offsetof(rt_object, position) == 16 && offsetof(rt_object, radius) == 28, "Object.position/radius");
RT_SA(offsetof(rt_object, normal) == 32 && offsetof(rt_object, size) == 48, "Object.normal/size");
RT_SA(offsetof(rt_object, material) == 64, "Object.material");
RT_SA(offsetof(rt_object, boundsMin) == 144 && offsetof(rt_object, boundsMax) == 160, "Object.bounds");
RT_SA(offsetof(rt_material, albedo) == 16 && offsetof(rt_material, metallic) == 28, "Material.albedo/metallic");
RT_SA(offsetof(rt_material, roughness) == 32 && offsetof(rt_material, diffuseStrength) == 36, "Material.roughness/diffuseStrength");
RT_SA(offsetof(rt_material, ior) == 40 && offsetof(rt_material, transparency) == 44, "Material.ior/transparency");
RT_SA(offsetof(rt_material, specular) == 48 && offsetof(rt_material, subsurfaceScatter) == 52, "Material.specular/sss");
RT_SA(offsetof(rt_material, subsurfaceColor) == 64 && offsetof(rt_material, scatterDistance) == 76, "Material.sssColor/scatterDistance");
RT_SA(offsetof(rt_light, position) == 16 && offsetof(rt_light, direction) == 32, "Light.position/direction");
RT_SA(offsetof(rt_light, color) == 48 && offsetof(rt_light, intensity) == 60, "Light.color/intensity");
RT_SA(offsetof(rt_light, radius) == 64 && offsetof(rt_light, samples) == 68, "Light.radius/samples");
RT_SA(offsetof(rt_light, shadowSoftness) == 72 && offsetof(rt_light, shadowType) == 76, "Light.shadowSoftness/shadowType");
RT_SA(offsetof(rt_light, pcfSamples) == 80 && offsetof(rt_light, lightSize) == 84, "Light.pcfSamples/lightSize");
RT_SA(offsetof(rt_light, angularRadius) == 88, "Light.angularRadius");
RT_SA(sizeof(rt_params) == 128, "rt_params is 128 B");
#undef RT_SA
#endif

#endif /* RT_MI355_H */
