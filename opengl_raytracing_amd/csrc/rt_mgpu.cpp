// rt_mgpu.cpp -- one frame on N GPUs of a node from ONE process and ONE host thread (include/rt_mi355.h, rt_mgpu_*).
//
// The reference is a single C++ process on one thread (/root/reference/src/main.cpp:3-7,
// ForwardShadingPipeline.cpp:129-271); a drop-in host cannot be asked to become one process per GPU.  Here one rt_context
// per device renders its interleaved row strips of the frame (pixels are independent: SURVEY.md 8(e)) and -- every device of
// an MI355X node is peer-mappable over xGMI -- its kernel STORES ITS ROWS STRAIGHT INTO DEVICE 0's full-frame surfaces
// (rt_render_into_image): no gather buffer, no pack / unpack pass, no collective; the 40 B/pixel leave the producing kernel
// as ordinary coalesced stores whose destination happens to be another GPU's HBM.  (bench.py --gpus N keeps the
// one-process-per-GPU torch.distributed / RCCL gather path the task's contract asks for and reports both.)
//
// Ordering, all by events, nothing blocks the host:
//   * frame k may only overwrite the root's surfaces once the root's stream has consumed frame k-1: every peer stream waits
//     for an event recorded on the root stream at the start of rt_mgpu_render;
//   * the root's stream waits for every peer's "my strips are stored" event, so whatever the caller enqueues on it next
//     (post passes, rt_mgpu_readback) sees the whole frame.
// Device ids may repeat (N "devices" that are all device 0): the N-way plan then executes in one process on a one-GPU box,
// which is how the tests rehearse N = 8.
#include <stdio.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "rt_mi355.h"

struct rt_mgpu {
    std::vector<rt_context *> ctx;
    std::vector<int> dev;
    std::vector<hipStream_t> stream;          // the contexts' own streams
    std::vector<hipEvent_t> done;             // device d's strips of the current frame are stored
    std::vector<hipEvent_t> t0, t1;           // timing of device d's share
    hipEvent_t frameStart = nullptr;          // root stream: the previous frame's consumers are behind this
    void *dColor = nullptr, *dPos = nullptr, *dNormal = nullptr;      // full-frame surfaces on device dev[0]
    size_t capPixels = 0;
    int W = 0, H = 0;
    int stripRows = 8;
    bool timed = false;
    std::string err;
};

namespace {
int mfail(rt_mgpu *m, int code, const std::string &what) {
    if (m) m->err = what;
    return code;
}
#define MG_HIP(m, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) return mfail(m, RT_ERR_HIP, std::string(#call ": ") + hipGetErrorString(e_)); \
    } while (0)
#define MG_RT(m, d, call)                                                                   \
    do {                                                                                    \
        int rc_ = (call);                                                                   \
        if (rc_) return mfail(m, rc_, std::string(#call " (device slot ") + std::to_string(d) + "): " + rt_last_error(m->ctx[d])); \
    } while (0)
}  // namespace

extern "C" {

int rt_mgpu_create(rt_mgpu **out, const int *deviceIds, int nDevices) {
    if (!out || !deviceIds || nDevices < 1 || nDevices > 64) return RT_ERR_INVALID_ARG;
    *out = nullptr;
    rt_mgpu *m = new (std::nothrow) rt_mgpu();
    if (!m) return RT_ERR_HIP;
    for (int d = 0; d < nDevices; d++) {
        rt_context *c = nullptr;
        int rc = rt_create(&c, deviceIds[d]);
        if (rc) {
            rt_mgpu_destroy(m);
            return rc;
        }
        m->ctx.push_back(c);
        m->dev.push_back(deviceIds[d]);
        void *s = nullptr;
        rt_context_stream(c, &s);
        m->stream.push_back((hipStream_t)s);
        hipEvent_t e = nullptr, a = nullptr, b = nullptr;
        if (hipSetDevice(deviceIds[d]) != hipSuccess || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess ||
            hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
            rt_mgpu_destroy(m);
            return RT_ERR_HIP;
        }
        m->done.push_back(e);
        m->t0.push_back(a);
        m->t1.push_back(b);
        // the peers' kernels store into the root's memory
        if (d > 0 && deviceIds[d] != deviceIds[0]) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, deviceIds[d], deviceIds[0]) != hipSuccess || !can) {
                rt_mgpu_destroy(m);
                return RT_ERR_NO_DEVICE;
            }
            const hipError_t pe = hipDeviceEnablePeerAccess(deviceIds[0], 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
                rt_mgpu_destroy(m);
                return RT_ERR_HIP;
            }
            (void)hipGetLastError();
        }
    }
    if (hipSetDevice(deviceIds[0]) != hipSuccess || hipEventCreateWithFlags(&m->frameStart, hipEventDisableTiming) != hipSuccess) {
        rt_mgpu_destroy(m);
        return RT_ERR_HIP;
    }
    *out = m;
    return RT_OK;
}

int rt_mgpu_destroy(rt_mgpu *m) {
    if (!m) return RT_ERR_INVALID_ARG;
    for (size_t d = 0; d < m->ctx.size(); d++) {
        (void)hipSetDevice(m->dev[d]);
        (void)rt_sync(m->ctx[d]);
    }
    for (size_t d = 0; d < m->ctx.size(); d++) {
        (void)hipSetDevice(m->dev[d]);
        if (d < m->done.size() && m->done[d]) (void)hipEventDestroy(m->done[d]);
        if (d < m->t0.size() && m->t0[d]) (void)hipEventDestroy(m->t0[d]);
        if (d < m->t1.size() && m->t1[d]) (void)hipEventDestroy(m->t1[d]);
        (void)rt_destroy(m->ctx[d]);
    }
    if (!m->dev.empty()) {
        (void)hipSetDevice(m->dev[0]);
        if (m->frameStart) (void)hipEventDestroy(m->frameStart);
        for (void *b : {m->dColor, m->dPos, m->dNormal})
            if (b) (void)hipFree(b);
    }
    delete m;
    return RT_OK;
}

int rt_mgpu_device_count(rt_mgpu *m) { return m ? (int)m->ctx.size() : RT_ERR_INVALID_ARG; }

int rt_mgpu_set_scene(rt_mgpu *m, const void *objects, int nObj, const void *lights, int nLt) {
    if (!m) return RT_ERR_INVALID_ARG;
    for (size_t d = 0; d < m->ctx.size(); d++) MG_RT(m, d, rt_set_scene(m->ctx[d], objects, nObj, lights, nLt));
    return RT_OK;
}

int rt_mgpu_set_noise(rt_mgpu *m, const uint8_t *r8, int w, int h) {
    if (!m) return RT_ERR_INVALID_ARG;
    for (size_t d = 0; d < m->ctx.size(); d++) MG_RT(m, d, rt_set_noise(m->ctx[d], r8, w, h));
    return RT_OK;
}

int rt_mgpu_set_skybox(rt_mgpu *m, const uint16_t *rgb16f, int size) {
    if (!m) return RT_ERR_INVALID_ARG;
    for (size_t d = 0; d < m->ctx.size(); d++) MG_RT(m, d, rt_set_skybox(m->ctx[d], rgb16f, size));
    return RT_OK;
}

int rt_mgpu_set_strip_rows(rt_mgpu *m, int stripRows) {
    if (!m || stripRows < 1) return RT_ERR_INVALID_ARG;
    m->stripRows = stripRows;
    return RT_OK;
}

int rt_mgpu_render(rt_mgpu *m, const rt_params *p) {
    if (!m || !p) return RT_ERR_INVALID_ARG;
    if (p->width <= 0 || p->height <= 0) return mfail(m, RT_ERR_INVALID_ARG, "width/height must be positive");
    if (p->x0 != 0 || p->y0 != 0 || p->regionW != p->width || p->regionH != p->height || p->stripCycleRows != 0 || p->stripCount != 1 ||
        p->stripRows != 1 || p->stripIndex != 0)
        return mfail(m, RT_ERR_INVALID_ARG, "rt_mgpu_render takes the parameters of the whole frame (the strips are its own business)");
    const int n = (int)m->ctx.size();
    const size_t npx = (size_t)p->width * p->height;
    MG_HIP(m, hipSetDevice(m->dev[0]));
    if (npx > m->capPixels) {
        for (int d = 0; d < n; d++) {
            MG_HIP(m, hipSetDevice(m->dev[d]));
            MG_RT(m, d, rt_sync(m->ctx[d]));
        }
        MG_HIP(m, hipSetDevice(m->dev[0]));
        for (void **b : {&m->dColor, &m->dPos, &m->dNormal}) {
            if (*b) MG_HIP(m, hipFree(*b));
            *b = nullptr;
        }
        m->capPixels = 0;
        MG_HIP(m, hipMalloc(&m->dColor, npx * 16));
        MG_HIP(m, hipMalloc(&m->dPos, npx * 16));
        MG_HIP(m, hipMalloc(&m->dNormal, npx * 8));
        m->capPixels = npx;
    }
    m->W = p->width;
    m->H = p->height;
    // everything the caller enqueued on the root stream so far (consumers of the previous frame) comes first
    MG_HIP(m, hipEventRecord(m->frameStart, m->stream[0]));
    for (int d = 0; d < n; d++) {
        const int rows = rt_strip_local_rows(p->height, m->stripRows, n, d);
        MG_HIP(m, hipSetDevice(m->dev[d]));
        if (d > 0) MG_HIP(m, hipStreamWaitEvent(m->stream[d], m->frameStart, 0));
        MG_HIP(m, hipEventRecord(m->t0[d], m->stream[d]));
        if (rows > 0) {
            rt_params q = *p;
            q.regionH = rows;
            q.stripRows = m->stripRows;
            q.stripCount = n;
            q.stripIndex = d;
            MG_RT(m, d, rt_render_into_image(m->ctx[d], &q, m->dColor, m->dPos, m->dNormal, nullptr));
        }
        MG_HIP(m, hipEventRecord(m->t1[d], m->stream[d]));
        if (d > 0) MG_HIP(m, hipEventRecord(m->done[d], m->stream[d]));
    }
    MG_HIP(m, hipSetDevice(m->dev[0]));
    for (int d = 1; d < n; d++) MG_HIP(m, hipStreamWaitEvent(m->stream[0], m->done[d], 0));
    m->timed = true;
    return RT_OK;
}

int rt_mgpu_sync(rt_mgpu *m) {
    if (!m) return RT_ERR_INVALID_ARG;
    // the root's stream is ordered behind every peer's stores
    MG_HIP(m, hipSetDevice(m->dev[0]));
    MG_RT(m, 0, rt_sync(m->ctx[0]));
    return RT_OK;
}

int rt_mgpu_get_surfaces(rt_mgpu *m, void **dColor, void **dPosition, void **dNormal, void **rootStream) {
    if (!m) return RT_ERR_INVALID_ARG;
    if (!m->dColor) return mfail(m, RT_ERR_NO_SURFACES, "no rendered frame");
    if (dColor) *dColor = m->dColor;
    if (dPosition) *dPosition = m->dPos;
    if (dNormal) *dNormal = m->dNormal;
    if (rootStream) *rootStream = (void *)m->stream[0];
    return RT_OK;
}

int rt_mgpu_readback(rt_mgpu *m, float *gColor, float *gPosition, uint16_t *gNormal) {
    if (!m) return RT_ERR_INVALID_ARG;
    if (!m->dColor || m->W <= 0) return mfail(m, RT_ERR_NO_SURFACES, "no rendered frame");
    int rc = rt_mgpu_sync(m);
    if (rc) return rc;
    const size_t npx = (size_t)m->W * m->H;
    if (gColor) MG_HIP(m, hipMemcpy(gColor, m->dColor, npx * 16, hipMemcpyDeviceToHost));
    if (gPosition) MG_HIP(m, hipMemcpy(gPosition, m->dPos, npx * 16, hipMemcpyDeviceToHost));
    if (gNormal) MG_HIP(m, hipMemcpy(gNormal, m->dNormal, npx * 8, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_mgpu_last_ms(rt_mgpu *m, float *perDeviceMs, int cap) {
    if (!m || !perDeviceMs) return RT_ERR_INVALID_ARG;
    if (!m->timed) return mfail(m, RT_ERR_NO_SURFACES, "no timed frame yet");
    const int n = (int)m->ctx.size() < cap ? (int)m->ctx.size() : cap;
    for (int d = 0; d < n; d++) {
        MG_HIP(m, hipSetDevice(m->dev[d]));
        MG_HIP(m, hipEventSynchronize(m->t1[d]));
        MG_HIP(m, hipEventElapsedTime(&perDeviceMs[d], m->t0[d], m->t1[d]));
    }
    return RT_OK;
}

const char *rt_mgpu_last_error(rt_mgpu *m) { return m ? m->err.c_str() : "NULL rt_mgpu"; }

}  // extern "C"
