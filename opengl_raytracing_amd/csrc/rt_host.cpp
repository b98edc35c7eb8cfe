// rt_host.cpp -- host-side feeders of the byte contract (no GPU involved): the pieces of
// the reference's C++ host that produce the bytes and uniforms the kernel trusts.
//   rt_generate_aabb   <- GenerateAABBForObject   /root/reference/src/SceneIO.h:75-104
//   rt_camera_vectors  <- Camera::UpdateVectors   /root/reference/src/Camera.h:26-34
//   rt_scene_parse     <- SceneIO::Load + ParseObject/ParseLight  SceneIO.h:108-122,145-186
//   rt_camera_matrices <- Camera::GetViewMatrix / GetProjectionMatrix  Camera.h:36-42
// glm is not available here; the few glm calls used there are restated in fp32
// (glm::normalize(v) = v * inversesqrt(dot(v,v)), dot = x*x + y*y + z*z, glm::radians(d) =
// d * 0.01745329251994329576923690768489f).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <sstream>
#include <string>

#include "rt_mi355.h"

namespace {

struct f3 { float x, y, z; };
inline f3 mk(float x, float y, float z) { f3 r = {x, y, z}; return r; }
inline f3 add(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline f3 sub(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline f3 mul(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline f3 crs(f3 a, f3 b) { return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline f3 nrm(f3 v) { float d = v.x * v.x + v.y * v.y + v.z * v.z; return mul(v, 1.0f / sqrtf(d)); }
inline f3 ld(const float *p) { return mk(p[0], p[1], p[2]); }
inline void st(float *p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

void aabb_one(rt_object *o) {
    if (o->type == 0) {   // sphere: centre +- radius (SceneIO.h:76-80)
        f3 p = ld(o->position), r = mk(o->radius, o->radius, o->radius);
        st(o->boundsMin, sub(p, r));
        st(o->boundsMax, add(p, r));
    } else if (o->type == 1) {   // plane (SceneIO.h:81-103)
        f3 n = ld(o->normal), right, forward;
        if (fabsf(n.y) > 0.9f) {
            right = mk(1, 0, 0);
            forward = mk(0, 0, 1);
        } else {
            right = nrm(crs(n, mk(0, 1, 0)));
            forward = nrm(crs(right, n));
        }
        f3 hx = mul(right, o->size[0] / 2.0f), hy = mul(forward, o->size[1] / 2.0f);
        f3 p = ld(o->position);
        f3 mn = sub(sub(p, hx), hy), mx = add(add(p, hx), hy);
        mn = add(mn, mul(n, 0.01f));   // zero-thickness box shifted 1 cm along the normal (:97-102)
        mx = add(mx, mul(n, 0.01f));
        st(o->boundsMin, mn);
        st(o->boundsMax, mx);
    }
}

void default_object(rt_object *o) {   // Object.h:16-18, Material.h:12-22
    memset(o, 0, sizeof *o);
    o->radius = 1.0f;
    o->normal[1] = 1.0f;
    o->size[0] = o->size[1] = 1.0f;
    o->material.type = 2;
    o->material.albedo[0] = o->material.albedo[1] = o->material.albedo[2] = 1.0f;
    o->material.roughness = 0.5f;
    o->material.diffuseStrength = 0.0f;   // indeterminate upstream (Material.h:16); defined as 0 here
    o->material.ior = 1.0f;
    o->material.specular = 0.5f;
    o->material.subsurfaceColor[0] = o->material.subsurfaceColor[1] = o->material.subsurfaceColor[2] = 1.0f;
    o->material.scatterDistance = 0.1f;
}

void default_light(rt_light *l) {   // Light.h:8-19
    memset(l, 0, sizeof *l);
    l->direction[1] = -1.0f;
    l->color[0] = l->color[1] = l->color[2] = 1.0f;
    l->intensity = 1.0f;
    l->radius = 0.5f;
    l->samples = 4;
    l->shadowSoftness = 1.0f;
    l->shadowType = 1;
    l->pcfSamples = 4;
    l->lightSize = 1.0f;
    l->angularRadius = 0.0f;   // indeterminate upstream (Light.h:19); dead in the shader
}

}  // namespace

extern "C" {

int rt_generate_aabb(void *objects, int n) {
    if (n < 0 || (n > 0 && !objects)) return RT_ERR_INVALID_ARG;
    rt_object *o = (rt_object *)objects;
    for (int i = 0; i < n; i++) aabb_one(&o[i]);
    return RT_OK;
}

int rt_camera_vectors(float yawDeg, float pitchDeg, float front[3], float right[3], float up[3]) {
    if (!front || !right || !up) return RT_ERR_INVALID_ARG;
    const float k = 0.01745329251994329576923690768489f;
    float yaw = yawDeg * k, pitch = pitchDeg * k;
    f3 f = nrm(mk(cosf(yaw) * cosf(pitch), sinf(pitch), sinf(yaw) * cosf(pitch)));
    f3 r = nrm(crs(f, mk(0, 1, 0)));
    f3 u = nrm(crs(r, f));
    st(front, f);
    st(right, r);
    st(up, u);
    return RT_OK;
}

int rt_camera_matrices(const float position[3], const float front[3], const float up[3], float fovDeg, float aspect,
                       float view[16], float projection[16]) {
    if (!position || !front || !up || !view || !projection) return RT_ERR_INVALID_ARG;
    // glm::lookAtRH(eye, center, up) (Camera.h:36-38), m[col*4 + row]
    const f3 eye = ld(position), center = add(eye, ld(front));
    const f3 f = nrm(sub(center, eye));
    const f3 s = nrm(crs(f, ld(up)));
    const f3 u = crs(s, f);
    auto dot3 = [](f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
    memset(view, 0, 16 * sizeof(float));
    view[0] = s.x; view[4] = s.y; view[8] = s.z;
    view[1] = u.x; view[5] = u.y; view[9] = u.z;
    view[2] = -f.x; view[6] = -f.y; view[10] = -f.z;
    view[12] = -dot3(s, eye); view[13] = -dot3(u, eye); view[14] = dot3(f, eye);
    view[15] = 1.0f;
    // glm::perspectiveRH_NO(radians(FOV), aspect, 0.1, 100) (Camera.h:40-42)
    const float fovy = fovDeg * 0.01745329251994329576923690768489f, zNear = 0.1f, zFar = 100.0f;
    const float tanHalf = tanf(fovy / 2.0f);
    memset(projection, 0, 16 * sizeof(float));
    projection[0] = 1.0f / (aspect * tanHalf);
    projection[5] = 1.0f / tanHalf;
    projection[10] = -(zFar + zNear) / (zFar - zNear);
    projection[11] = -1.0f;
    projection[14] = -(2.0f * zFar * zNear) / (zFar - zNear);
    return RT_OK;
}

int rt_scene_parse(const char *text, void *objects, int maxObj, int *nObj, void *lights, int maxLt, int *nLt) {
    if (!text || !nObj || !nLt || maxObj < 0 || maxLt < 0) return RT_ERR_INVALID_ARG;
    rt_object *objs = (rt_object *)objects;
    rt_light *lts = (rt_light *)lights;
    int no = 0, nl = 0;
    std::istringstream all(text);
    std::string line;
    while (std::getline(all, line)) {
        std::istringstream iss(line);
        std::string kind;
        iss >> kind;
        if (kind == "OBJECT") {   // ParseObject (SceneIO.h:145-170)
            rt_object o;
            default_object(&o);
            std::string typeStr, name;
            iss >> typeStr >> name;
            iss >> o.position[0] >> o.position[1] >> o.position[2];
            o.type = (typeStr == "PLANE") ? 1 : 0;   // StringToObjectType: unknown -> SPHERE
            iss >> o.radius >> o.normal[0] >> o.normal[1] >> o.normal[2] >> o.size[0] >> o.size[1];
            int matType = 0;
            iss >> matType;
            o.material.type = matType;
            iss >> o.material.albedo[0] >> o.material.albedo[1] >> o.material.albedo[2] >> o.material.metallic >>
                o.material.roughness >> o.material.ior >> o.material.transparency >> o.material.specular;
            aabb_one(&o);
            if (no < maxObj && objs) objs[no] = o;
            no++;
        } else if (kind == "LIGHT") {   // ParseLight (SceneIO.h:172-186)
            rt_light l;
            default_light(&l);
            std::string typeStr, name;
            iss >> typeStr >> name;
            l.type = (typeStr == "DIRECTIONAL") ? 1 : (typeStr == "AREA") ? 2 : 0;
            iss >> l.position[0] >> l.position[1] >> l.position[2] >> l.direction[0] >> l.direction[1] >>
                l.direction[2] >> l.color[0] >> l.color[1] >> l.color[2] >> l.intensity >> l.radius >> l.samples;
            if (nl < maxLt && lts) lts[nl] = l;
            nl++;
        }
    }
    *nObj = no;
    *nLt = nl;
    if (no > maxObj || nl > maxLt) return RT_ERR_TOO_LARGE;
    return RT_OK;
}

int rt_scene_write(const void *objects, int nObj, const void *lights, int nLt, const char *const *objNames,
                   const char *const *lightNames, char *out, size_t cap, size_t *needed) {
    if (nObj < 0 || nLt < 0 || (nObj > 0 && !objects) || (nLt > 0 && !lights) || !needed) return RT_ERR_INVALID_ARG;
    const rt_object *objs = (const rt_object *)objects;
    const rt_light *lts = (const rt_light *)lights;
    std::ostringstream file;   // same default float formatting as the std::ofstream of SceneIO::Save
    for (int i = 0; i < nObj; i++) {   // WriteObjectParams (SceneIO.h:50-64)
        const rt_object &o = objs[i];
        std::string name = (objNames && objNames[i]) ? objNames[i] : ("Object" + std::to_string(i));
        file << "OBJECT " << (o.type == 0 ? "SPHERE" : o.type == 1 ? "PLANE" : "UNKNOWN");
        file << " " << name << " " << o.position[0] << " " << o.position[1] << " " << o.position[2] << " " << o.radius << " "
             << o.normal[0] << " " << o.normal[1] << " " << o.normal[2] << " " << o.size[0] << " " << o.size[1] << " "
             << o.material.type << " " << o.material.albedo[0] << " " << o.material.albedo[1] << " " << o.material.albedo[2]
             << " " << o.material.metallic << " " << o.material.roughness << " " << o.material.ior << " "
             << o.material.transparency << " " << o.material.specular;
        file << "\n";
    }
    for (int i = 0; i < nLt; i++) {    // WriteLightParams (SceneIO.h:66-73)
        const rt_light &l = lts[i];
        std::string name = (lightNames && lightNames[i]) ? lightNames[i] : ("Light" + std::to_string(i));
        file << "LIGHT " << (l.type == 1 ? "DIRECTIONAL" : l.type == 2 ? "AREA" : "POINT");
        file << " " << name << " " << l.position[0] << " " << l.position[1] << " " << l.position[2] << " " << l.direction[0]
             << " " << l.direction[1] << " " << l.direction[2] << " " << l.color[0] << " " << l.color[1] << " " << l.color[2]
             << " " << l.intensity << " " << l.radius << " " << l.samples;
        file << "\n";
    }
    const std::string text = file.str();
    *needed = text.size() + 1;
    if (!out || cap < *needed) return out ? RT_ERR_TOO_LARGE : RT_OK;
    memcpy(out, text.c_str(), *needed);
    return RT_OK;
}

int rt_taa_jitter(int frameCount, int width, int height, float *jitterX, float *jitterY) {
    if (!jitterX || !jitterY || width <= 0 || height <= 0) return RT_ERR_INVALID_ARG;
    // haltonSequence of /root/reference/src/global.cpp:41-51 (note its floor(i / base) on ints)
    auto halton = [](int index, int base) {
        float result = 0.0f, f = 1.0f / (float)base;
        int i = index;
        while (i > 0) { result += f * (float)(i % base); i = (int)floorf((float)(i / base)); f /= (float)base; }
        return result;
    };
    *jitterX = halton(frameCount % 8, 2) * 0.5f / (float)width;    // ForwardShadingPipeline.cpp:241
    *jitterY = halton(frameCount % 8, 3) * 0.5f / (float)height;   // :242
    return RT_OK;
}

}  // extern "C"
