// host_swap.cpp -- what the reference's C++ host does around its glDispatchCompute
// (/root/reference/src/ForwardShadingPipeline.cpp:155-182), written against the C ABI only:
// fill std::vector<rt_object>/<rt_light> (the reference's Object/Light bytes), set the camera
// uniforms, render, read the three surfaces back.  Prints an FNV-1a hash per surface so the
// Python-side test can check that this path and the ctypes path produce the same bytes.
//   g++ -std=c++17 -I include examples/host_swap.cpp -L opengl_raytracing_amd -lrt_mi355 -o host_swap
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rt_mi355.h"

static uint64_t fnv1a(const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 320, H = argc > 2 ? atoi(argv[2]) : 180;
    // a default.scene-like set-up: three spheres on a ground plane, three lights
    const char *scene =
        "OBJECT SPHERE Metal -2.5 0.5 -5 1 0 0 0 0 0 0 0.9 0.85 0.8 1 0.2 1 0 0.5\n"
        "OBJECT SPHERE Glass 0 0.5 -5 0.8 0 0 0 0 0 1 0.9 0.95 1 0 0.05 1.5 0.95 0.5\n"
        "OBJECT SPHERE Plastic 2.5 0.5 -5 1 0 0 0 0 0 2 0.2 0.5 0.8 0 0.5 1 0 0.6\n"
        "OBJECT PLANE Ground 0 -1 -5 0 0 1 0 10 10 2 0.8 0.8 0.8 0 0.6 1 0 0.5\n"
        "LIGHT DIRECTIONAL Sun 0 5 0 0.5 -1 -0.5 1 1 1 3 0 1\n"
        "LIGHT AREA Panel 0 3.5 0 0 1 0 1 1 0.9 5 0.5 16\n"
        "LIGHT POINT Bulb 0 2.5 -3 0 0 0 1 0.8 0.7 8 1.5 0\n";
    std::vector<rt_object> objects(16);
    std::vector<rt_light> lights(8);
    int nObj = 0, nLt = 0;
    if (rt_scene_parse(scene, objects.data(), (int)objects.size(), &nObj, lights.data(), (int)lights.size(), &nLt) != RT_OK) return 2;
    objects.resize(nObj);
    lights.resize(nLt);

    rt_context *rt = nullptr;
    int rc = rt_create(&rt, 0);
    if (rc != RT_OK) { fprintf(stderr, "rt_create failed: %d (no CPU fallback)\n", rc); return 3; }
    rc = rt_set_scene(rt, objects.data(), nObj, lights.data(), nLt);       // ssbo.update(); lightSSBO.update()
    if (rc != RT_OK) { fprintf(stderr, "%s\n", rt_last_error(rt)); return 4; }

    rt_params p;
    memset(&p, 0, sizeof p);
    float front[3], right[3], up[3];
    rt_camera_vectors(-90.0f, 0.0f, front, right, up);                      // Camera::UpdateVectors
    const float pos[3] = {0.0f, 1.0f, 3.0f};
    memcpy(p.camPos, pos, 12); memcpy(p.camDir, front, 12); memcpy(p.camUp, up, 12); memcpy(p.camRight, right, 12);
    p.fovDeg = 45.0f; p.focalLength = 1.0f; p.maxRayDistance = 114514.0f;
    p.noiseScale[0] = p.noiseScale[1] = 1.0f / 1024.0f;
    p.frameCount = 0; p.useSkybox = 0; p.maxRayDepth = 3;
    p.width = W; p.height = H; p.regionW = W; p.regionH = H;
    p.stripRows = 1; p.stripCount = 1; p.stripIndex = 0;
    rc = rt_render(rt, &p);                                                // glDispatchCompute + barrier
    if (rc != RT_OK) { fprintf(stderr, "%s\n", rt_last_error(rt)); return 5; }
    float ms = 0;
    rt_last_kernel_ms(rt, &ms);
    std::vector<float> color((size_t)W * H * 4), position((size_t)W * H * 4);
    std::vector<uint16_t> normal((size_t)W * H * 4);
    rc = rt_readback(rt, color.data(), position.data(), normal.data());    // glGetTexImage x3
    if (rc != RT_OK) { fprintf(stderr, "%s\n", rt_last_error(rt)); return 6; }
    uint64_t rays = 0;
    rt_count_rays(rt, &p, &rays);
    printf("{\"width\": %d, \"height\": %d, \"kernel_ms\": %.4f, \"rays\": %llu, \"color\": \"%016llx\", "
           "\"position\": \"%016llx\", \"normal\": \"%016llx\"}\n",
           W, H, ms, (unsigned long long)rays, (unsigned long long)fnv1a(color.data(), color.size() * 4),
           (unsigned long long)fnv1a(position.data(), position.size() * 4),
           (unsigned long long)fnv1a(normal.data(), normal.size() * 2));
    rt_destroy(rt);
    return 0;
}
