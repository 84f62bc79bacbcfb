// ORACLE — test infrastructure only.  Never linked, imported or called by the product path
// (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/).
//
// Scalar float32 CPU restatement of the reference integrator
//   /root/reference/Raytracing/shader.comp:104-125 (RayIntersectsTriangle), 132-151 (barycentric),
//   153-158 (hash13), 171-184 (intersectAABB), 186-191 (fresnelReflectance), 193-241 (sampleGGX),
//   243-290 (lineCast), 292-537 (main)
// running directly on the reference's SSBO byte layouts (shader.comp:14-83).  One function per
// shader function, same evaluation order, no FMA contraction (build with -ffp-contract=off).
//
// GLSL built-ins are written out per the GLSL 4.30 spec (section 8): min(x,y) = y<x?y:x,
// max(x,y) = x<y?y:x, clamp = min(max(x,lo),hi), fract = x-floor(x), mix = x*(1-a)+y*a,
// reflect = I-2*dot(N,I)*N, refract per spec, length = sqrt(dot), normalize = v*(1/sqrt(dot(v,v))),
// dot summed left to right, mat4*vec4 column-major summed left to right.
// sin/cos/pow are driver-dependent in GLSL; math_mode selects:
//   0 = libm  (sinf, cosf, powf)
//   1 = portable (DESIGN.md "Portable transcendentals": Cody-Waite + Cephes-style polynomials using
//       only + - * so that any IEEE-754 implementation reproduces them bit for bit; pow(x,5)=((x*x)*(x*x))*x,
//       pow(x,2)=x*x)
//
// Pinning: PARITY UNPINNED by the reference — it ships no tests, golden images or known-answer
// vectors and shader.comp cannot be executed in this image (no GLSL compiler,
// no GL context).  The restatement is cross-checked by an independent numpy re-derivation of
// hash13 / ray-gen / lineCast (tests/test_oracle_rt.py) and by hand-computed known answers.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct vec3 { float x, y, z; };
struct vec2 { float x, y; };
struct vec4 { float x, y, z, w; };

inline vec3 V(float a, float b, float c) { return {a, b, c}; }
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator+(vec3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator-(float s, vec3 a) { return {s - a.x, s - a.y, s - a.z}; }

inline float gmin(float x, float y) { return y < x ? y : x; }
inline float gmax(float x, float y) { return x < y ? y : x; }
inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
inline vec3 gmin(vec3 a, vec3 b) { return {gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)}; }
inline vec3 gmax(vec3 a, vec3 b) { return {gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 x, vec3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
inline float length(vec3 a) { return sqrtf(dot(a, a)); }
inline float distance(vec3 a, vec3 b) { return length(a - b); }
inline vec3 normalize(vec3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
inline float fract(float x) { return x - floorf(x); }
inline vec3 fract(vec3 a) { return {fract(a.x), fract(a.y), fract(a.z)}; }
inline vec3 reflect(vec3 I, vec3 N) { return I - (2.0f * dot(N, I)) * N; }
inline vec3 refract(vec3 I, vec3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return V(0.f, 0.f, 0.f);
    return eta * I - (eta * d + sqrtf(k)) * N;
}
inline vec3 mix(vec3 x, vec3 y, float a) { return x * (1.0f - a) + y * a; }

// ---- reference SSBO record layouts (read with memcpy-free casts; buffers are 4-byte aligned) ----
struct Vertex { vec3 position; float p1; vec3 normal; float p2; vec3 color; float p3; vec2 texUV; vec2 p4; vec4 tangent; };  // VBO.h:9-17
struct MaterialData {                                              // model.h:8-39
    vec4 baseColorFactor;
    uint64_t baseColorTexture; int32_t hasBaseTexture; float transmissionFactor;
    uint64_t normalTexture; int32_t hasNormalTexture; float metallicFactor;
    uint64_t metallicRoughnessTexture; int32_t hasMetallicRoughnessTexture; float roughnessFactor;
    float ior; float pad1[3];
};
struct BoundingBox { vec3 minLoc; float pad1; vec3 maxLoc; uint32_t isLeaf; int32_t children[4]; };  // model.h:130-142
struct Triangle {                                                  // model.h:41-70
    uint32_t indices[3]; uint32_t p0; vec3 p1; float q1; vec3 p2; float q2; vec3 p3; float q3;
    vec3 centroidLoc; float q4; vec3 minBounds; float q5; vec3 maxBounds; uint32_t meshIndex;
    uint32_t materialIndex; uint32_t pad1[2]; uint32_t tail;
};
struct Light { vec3 loc; float strength; vec3 dir; uint32_t lightType; vec3 color; float pad1; };  // light.h:14-23
struct mat4 { float m[4][4]; };  // [col][row]
static_assert(sizeof(Vertex) == 80 && sizeof(MaterialData) == 80 && sizeof(BoundingBox) == 48 &&
              sizeof(Triangle) == 128 && sizeof(Light) == 48 && sizeof(mat4) == 64, "std430 layout");

inline vec3 mulPoint(const mat4& M, vec3 v, float w) {  // vec3(M * vec4(v, w))
    return {M.m[0][0] * v.x + M.m[1][0] * v.y + M.m[2][0] * v.z + M.m[3][0] * w,
            M.m[0][1] * v.x + M.m[1][1] * v.y + M.m[2][1] * v.z + M.m[3][1] * w,
            M.m[0][2] * v.x + M.m[1][2] * v.y + M.m[2][2] * v.z + M.m[3][2] * w};
}

}  // namespace

extern "C" {

struct RtRefTexture { const uint8_t* rgba8; uint32_t width, height; };

struct RtRefScene {
    const void* vertices;  uint64_t n_vertices;     // slot 0
    const void* matrices;  uint64_t n_matrices;     // slot 1
    const void* materials; uint64_t n_materials;    // slot 2
    const void* nodes;     uint64_t n_nodes;        // slot 3
    const void* triangles; uint64_t n_triangles;    // slot 4
    const void* lights;    uint64_t n_lights;       // slot 5
    const RtRefTexture* textures; uint64_t n_textures;   // handle h (1-based) -> textures[h-1]
};

struct RtRefParams {
    int32_t width, height;
    int32_t numLights, hasSun;
    float camPos[3], camForward[3], camRight[3], camUp[3];
    float aspect, fov;
    int32_t samplesPerPixel, maxBounces;
    int32_t spp_begin, spp_end;     // sample range actually traced (multi-GPU partition checks); full = [0, spp)
    int32_t x0, y0, x1, y1;         // pixel region [x0,x1) x [y0,y1); outside stays untouched
    int32_t math_mode;              // 0 libm, 1 portable
    int32_t write_aovs;             // write albedo/normal/depth (the rank owning r = 0)
};

struct RtRefStats {
    uint64_t rays_primary, rays_bounce, rays_shadow;
    uint64_t node_visits, tri_tests;
};

}  // extern "C"

namespace {

struct Ctx {
    const Vertex* vertices; const mat4* meshProj; const MaterialData* materialList;
    const BoundingBox* boundingBoxList; const Triangle* triangleList; const Light* lights;
    const RtRefTexture* textures; uint64_t n_textures;
    int math_mode;
};

// ---- portable transcendentals.  Only + - * and float<->int conversion. ----
inline void portable_sincos(float phi, float* s_out, float* c_out) {
    int k = (int)(phi * 0.636619772f + 0.5f);       // nearest multiple of pi/2 (phi >= 0)
    float kf = (float)k;
    float r = phi - kf * 1.5703125f;                // Cody-Waite 3-term pi/2
    r = r - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.54978995489188216e-8f;
    float z = r * r;
    float s = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float c = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    switch (k & 3) {
        case 0: *s_out = s;  *c_out = c;  break;
        case 1: *s_out = c;  *c_out = -s; break;
        case 2: *s_out = -s; *c_out = -c; break;
        default: *s_out = -c; *c_out = s; break;
    }
}
inline float pow5(const Ctx& C, float x) { return C.math_mode ? ((x * x) * (x * x)) * x : powf(x, 5.0f); }
inline float pow2(const Ctx& C, float x) { return C.math_mode ? x * x : powf(x, 2.0f); }

// texture(sampler2D(handle), uv).  GL 4.3 section 8.14: LINEAR min/mag, REPEAT wrap, level 0, RGBA8 unorm
// (Texture.cpp:5-34).  Spec arithmetic written out in float32; drivers use fixed-point weights, so
// this is "parity unpinned" like every other built-in.
inline vec4 texture2D(const Ctx& C, uint64_t handle, vec2 uv) {
    if (handle == 0 || handle > C.n_textures) return {0.f, 0.f, 0.f, 1.f};
    const RtRefTexture& T = C.textures[handle - 1];
    float fu = uv.x * (float)T.width - 0.5f, fv = uv.y * (float)T.height - 0.5f;
    float iu = floorf(fu), iv = floorf(fv);
    float a = fu - iu, b = fv - iv;
    int64_t w = T.width, h = T.height;
    int64_t i0 = (int64_t)iu % w; if (i0 < 0) i0 += w;
    int64_t j0 = (int64_t)iv % h; if (j0 < 0) j0 += h;
    int64_t i1 = (i0 + 1) % w, j1 = (j0 + 1) % h;
    auto tex = [&](int64_t i, int64_t j, int c) { return (float)T.rgba8[(j * w + i) * 4 + c] / 255.0f; };
    float o[4];
    for (int c = 0; c < 4; c++) {
        float t00 = tex(i0, j0, c), t10 = tex(i1, j0, c), t01 = tex(i0, j1, c), t11 = tex(i1, j1, c);
        o[c] = ((1.0f - a) * (1.0f - b)) * t00 + (a * (1.0f - b)) * t10 + ((1.0f - a) * b) * t01 + (a * b) * t11;
    }
    return {o[0], o[1], o[2], o[3]};
}

// shader.comp:104-125
inline bool RayIntersectsTriangle(vec3 orig, vec3 dir, vec3 v0, vec3 v1, vec3 v2, vec3* hitLoc) {
    vec3 e1 = v1 - v0;
    vec3 e2 = v2 - v0;
    vec3 p = cross(dir, e2);
    float det = dot(e1, p);
    if (fabsf(det) < 1e-8f) return false;
    float invDet = 1.0f / det;
    vec3 s = orig - v0;
    float u = dot(s, p) * invDet;
    if (u < 0.0f || u > 1.0f) return false;
    vec3 q = cross(s, e1);
    float v = dot(dir, q) * invDet;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = dot(e2, q) * invDet;
    if (t < 1e-8f) return false;
    *hitLoc = orig + dir * t;
    return true;
}

struct Hit { vec3 hitLoc; Triangle triangle; int32_t triIndex; };

// shader.comp:132-151
inline vec3 barycentric(const Hit& hit) {
    vec3 p = hit.hitLoc, a = hit.triangle.p1, b = hit.triangle.p2, c = hit.triangle.p3;
    vec3 v0 = b - a, v1 = c - a, v2 = p - a;
    float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
    float denom = d00 * d11 - d01 * d01;
    float v = (d11 * d20 - d01 * d21) / denom;
    float w = (d00 * d21 - d01 * d20) / denom;
    float u = 1.0f - v - w;
    return {u, v, w};
}

// shader.comp:153-158
inline float hash13(vec3 p) {
    p = fract(p * 0.1031f);
    p = p + dot(p, V(p.y, p.z, p.x) + 33.33f);
    return fract((p.x + p.y) * p.z);
}

// shader.comp:171-184
inline bool intersectAABB(vec3 rayOrigin, vec3 rayDir, vec3 aabbMin, vec3 aabbMax) {
    vec3 invDir = V(1.0f / rayDir.x, 1.0f / rayDir.y, 1.0f / rayDir.z);
    vec3 t0 = (aabbMin - rayOrigin) * invDir;
    vec3 t1 = (aabbMax - rayOrigin) * invDir;
    vec3 tmin = gmin(t0, t1);
    vec3 tmax = gmax(t0, t1);
    float tEnter = gmax(gmax(tmin.x, tmin.y), tmin.z);
    float tExit = gmin(gmin(tmax.x, tmax.y), tmax.z);
    return tExit >= tEnter && tExit >= 0.0f;
}

// shader.comp:186-191
inline float fresnelReflectance(const Ctx& C, vec3 I, vec3 N, float ior1, float ior2) {
    float cosi = gclamp(dot(-I, N), -1.0f, 1.0f);
    float r0 = (ior1 - ior2) / (ior1 + ior2);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow5(C, 1.0f - cosi);
}

// shader.comp:193-241
inline vec3 sampleGGX(const Ctx& C, vec3 N, float rand1, float rand2, float alpha) {
    if (alpha == 0.f) return N;
    rand1 = gclamp(rand1, 0.0001f, 0.9999f);
    rand2 = gclamp(rand2, 0.0001f, 0.9999f);
    alpha = gmax(alpha, 0.0001f);
    float a2 = alpha * alpha;
    float phi = (2.0f * 3.141592653589793f) * rand1;
    float denom = 1.0f + (a2 - 1.0f) * rand2;
    denom = gmax(denom, 0.000001f);
    float cosTheta = sqrtf((1.0f - rand2) / denom);
    cosTheta = gclamp(cosTheta, -1.0f, 1.0f);
    float sinTheta = sqrtf(gmax(0.0f, 1.0f - cosTheta * cosTheta));
    float sp, cp;
    if (C.math_mode) portable_sincos(phi, &sp, &cp); else { sp = sinf(phi); cp = cosf(phi); }
    vec3 Ht = V(sinTheta * cp, sinTheta * sp, cosTheta);
    vec3 T;
    if (fabsf(N.z) < 0.999f) T = normalize(cross(V(0.f, 0.f, 1.f), N));
    else T = normalize(cross(V(1.f, 0.f, 0.f), N));
    vec3 B = cross(N, T);
    vec3 H = T * Ht.x + B * Ht.y + N * Ht.z;
    float lenH = length(H);
    if (lenH < 1e-6f) return N;
    return H / lenH;
}

// shader.comp:243-290
inline bool lineCast(const Ctx& C, vec3 startPos, vec3 startDir, Hit* hit, RtRefStats* st) {
    bool didHit = false;
    float closestDist = 3.40282347e+38f;
    const uint32_t stackSize = 64;
    uint32_t stack[stackSize];
    uint32_t stackPtr = 0;
    stack[stackPtr++] = 0;
    while (stackPtr > 0) {
        uint32_t nodeIndex = stack[--stackPtr];
        const BoundingBox& node = C.boundingBoxList[nodeIndex];
        st->node_visits++;
        if (intersectAABB(startPos, startDir, node.minLoc, node.maxLoc)) {
            if (node.isLeaf == 1) {
                for (int i = 0; i < 4; i++) {
                    if (node.children[i] != -1) {
                        const Triangle& triangle = C.triangleList[node.children[i]];
                        st->tri_tests++;
                        const mat4& M = C.meshProj[triangle.meshIndex];
                        vec3 p1 = mulPoint(M, C.vertices[triangle.indices[0]].position, 1.f);
                        vec3 p2 = mulPoint(M, C.vertices[triangle.indices[1]].position, 1.f);
                        vec3 p3 = mulPoint(M, C.vertices[triangle.indices[2]].position, 1.f);
                        vec3 _hitLoc;
                        if (RayIntersectsTriangle(startPos, startDir, p1, p2, p3, &_hitLoc)) {
                            float newDist = dot(startPos - _hitLoc, startPos - _hitLoc);
                            if (newDist < closestDist) {
                                didHit = true;
                                hit->hitLoc = _hitLoc;
                                closestDist = newDist;
                                hit->triangle = triangle;
                                hit->triIndex = node.children[i];
                            }
                        }
                    }
                }
            } else {
                if (node.children[0] != -1 && stackPtr < 64) stack[stackPtr++] = (uint32_t)node.children[0];
                if (node.children[1] != -1 && stackPtr < 64) stack[stackPtr++] = (uint32_t)node.children[1];
            }
        }
    }
    return didHit;
}

struct Ray { vec3 loc, dir, strength; bool refracting, isActive; };

// shader.comp:292-537 for one pixel.  prim_tri (optional) receives the sample-0 bounce-0 triangle id.
void shadePixel(const Ctx& C, const RtRefParams& P, float tanHalfFov, int px_, int py_,
                float* color_out, float* albedo_out, float* normal_out, float* depth_out,
                int32_t* prim_tri, float* prim_hit, RtRefStats* st) {
    const int pixel_x = px_, pixel_y = py_;
    const vec3 camPos = V(P.camPos[0], P.camPos[1], P.camPos[2]);
    const vec3 camForward = V(P.camForward[0], P.camForward[1], P.camForward[2]);
    const vec3 camRight = V(P.camRight[0], P.camRight[1], P.camRight[2]);
    const vec3 camUp = V(P.camUp[0], P.camUp[1], P.camUp[2]);
    const int samplesPerPixel = P.samplesPerPixel, maxBounces = P.maxBounces;

    vec3 albedoData = V(0, 0, 0), firstHitNormalData = V(0, 0, 0);
    float firstHitDepthData = 0.f;
    const vec3 ambientColor = P.hasSun ? V(0.4431372549f, 0.73725490196f, 1.f) : V(0.f, 0.f, 0.f);
    vec3 color = V(0, 0, 0);
    const float rayStrengthCutoff = .01f / (float)samplesPerPixel;
    const float initialRayStrength = 1.f / (float)samplesPerPixel;
    float sqrtActiveRaysNum = sqrtf((float)samplesPerPixel);
    int N = (int)ceilf(sqrtActiveRaysNum);
    if (prim_tri) *prim_tri = -1;

    for (int r = P.spp_begin; r < P.spp_end; ++r) {
        int gx = r % N, gy = r / N;
        vec2 jitter = {((float)gx + 0.5f) / (float)N, ((float)gy + 0.5f) / (float)N};
        vec2 uv = {((float)pixel_x + jitter.x) / (float)P.width * 2.0f - 1.0f,
                   ((float)pixel_y + jitter.y) / (float)P.height * 2.0f - 1.0f};
        uv.x *= P.aspect;
        float px = uv.x * tanHalfFov;
        float py = uv.y * tanHalfFov;
        vec3 rayDir = (camForward + px * camRight + py * camUp);
        Ray ray{camPos, rayDir, V(initialRayStrength, initialRayStrength, initialRayStrength), false, true};
        for (int bounce = 0; bounce < maxBounces; ++bounce) {
            if (!ray.isActive) continue;
            Hit hit;
            if (bounce == 0) st->rays_primary++; else st->rays_bounce++;
            if (!lineCast(C, ray.loc, ray.dir, &hit, st)) {
                color = color + ambientColor * ray.strength;
                ray.isActive = false;
                continue;
            }
            if (r == 0 && bounce == 0) {
                firstHitDepthData = distance(camPos, hit.hitLoc);
                if (prim_tri) *prim_tri = hit.triIndex;
                if (prim_hit) { prim_hit[0] = hit.hitLoc.x; prim_hit[1] = hit.hitLoc.y; prim_hit[2] = hit.hitLoc.z; }
            }
            uint32_t i1 = hit.triangle.indices[0], i2 = hit.triangle.indices[1], i3 = hit.triangle.indices[2];
            vec3 bary = barycentric(hit);
            const mat4& M = C.meshProj[hit.triangle.meshIndex];
            vec3 n1 = mulPoint(M, C.vertices[i1].normal, 0.f);
            vec3 n2 = mulPoint(M, C.vertices[i2].normal, 0.f);
            vec3 n3 = mulPoint(M, C.vertices[i3].normal, 0.f);
            vec3 hitNormal = (n1 * bary.x + n2 * bary.y + n3 * bary.z);
            vec2 uv1 = C.vertices[i1].texUV, uv2 = C.vertices[i2].texUV, uv3 = C.vertices[i3].texUV;
            vec2 hitUV = {uv1.x * bary.x + uv2.x * bary.y + uv3.x * bary.z, uv1.y * bary.x + uv2.y * bary.y + uv3.y * bary.z};
            const MaterialData& material = C.materialList[hit.triangle.materialIndex];

            float metallicFactor, roughnessFactor;
            if (material.hasMetallicRoughnessTexture == 1) {
                vec4 t = texture2D(C, material.metallicRoughnessTexture, hitUV);
                metallicFactor = t.x; roughnessFactor = t.y;
            } else { metallicFactor = material.metallicFactor; roughnessFactor = material.roughnessFactor; }

            vec3 normal;
            if (material.hasNormalTexture == 1) {
                vec4 t1 = C.vertices[i1].tangent, t2 = C.vertices[i2].tangent, t3 = C.vertices[i3].tangent;
                vec3 tangent = (V(t1.x, t1.y, t1.z) * bary.x + V(t2.x, t2.y, t2.z) * bary.y + V(t3.x, t3.y, t3.z) * bary.z);
                float handedness = t1.w * bary.x + t2.w * bary.y + t3.w * bary.z;
                vec4 nt = texture2D(C, material.normalTexture, hitUV);
                vec3 normalTex = V(nt.x, nt.y, nt.z) * 2.0f - V(1.f, 1.f, 1.f);
                float normalStrength = 1.f;
                normalTex.x *= normalStrength; normalTex.y *= normalStrength;
                vec3 bitangent = cross(hitNormal, tangent) * handedness;
                normal = tangent * normalTex.x + bitangent * normalTex.y + hitNormal * normalTex.z;  // mat3(T,B,N) * v
            } else normal = hitNormal;

            if (r == 0 && bounce == 0) firstHitNormalData = normal;

            bool shouldReflect = true;
            if (material.transmissionFactor > 0) {
                float iorAir = 1.f;
                float reflectionStrength = fresnelReflectance(C, ray.dir, normal, iorAir, material.ior);
                if (!ray.refracting) {
                    float rand = hash13(V((float)pixel_x * 2.3f, (float)pixel_y * 2.3f, (float)(r * bounce) * 18.f) * ray.dir + ray.loc);
                    if (rand > reflectionStrength) {
                        ray.dir = refract(ray.dir, normal, iorAir / material.ior);
                        ray.loc = hit.hitLoc - normal * 1e-5f;
                        ray.strength = ray.strength * material.transmissionFactor;
                        ray.refracting = true;
                        shouldReflect = false;
                    }
                } else {
                    vec3 newLoc = hit.hitLoc + normal * 1e-5f;
                    vec3 newDir = refract(ray.dir, -normal, material.ior / iorAir);
                    if (length(newDir) == 0.f) {
                        ray.dir = reflect(ray.dir, -normal);
                        ray.loc = hit.hitLoc - normal * 1e-5f;
                        ray.refracting = true;
                    } else {
                        ray.loc = newLoc;
                        ray.dir = newDir;
                        ray.refracting = false;
                    }
                    shouldReflect = false;
                }
            }
            if (shouldReflect) {
                float rb = (float)(r * bounce);
                float h1 = hash13(V((float)pixel_x, (float)pixel_y, rb) * ray.dir + ray.loc);
                float h2 = hash13(V((float)pixel_x * 37.f, (float)pixel_y * 37.f, rb * 37.f) * ray.dir + ray.loc);
                vec3 newDir = sampleGGX(C, normal, h1, h2, roughnessFactor);
                ray.dir = reflect(ray.dir, newDir);
                ray.loc = hit.hitLoc + (normal * 1e-5f);
            }

            vec3 baseColor = V(material.baseColorFactor.x, material.baseColorFactor.y, material.baseColorFactor.z);
            if (material.hasBaseTexture == 1) {
                vec4 t = texture2D(C, material.baseColorTexture, hitUV);
                baseColor = V(t.x, t.y, t.z);
            }
            vec3 viewDir = normalize(camPos - hit.hitLoc);
            float dielectricF0 = .04f;
            if (material.transmissionFactor > 0) dielectricF0 = pow2(C, (material.ior - 1.0f) / (material.ior + 1.0f));
            vec3 F0 = mix(V(dielectricF0, dielectricF0, dielectricF0), baseColor, metallicFactor);
            float alpha = roughnessFactor * roughnessFactor;
            float PI = 3.14159265359f;
            vec3 diffuseColor = baseColor * (1.0f - metallicFactor);
            if (r == 0 && bounce == 0) albedoData = diffuseColor;

            for (int l = 0; l < P.numLights; l++) {
                const Light& light = C.lights[l];
                vec3 dirToLight = V(0.f, 0.f, 0.f);
                float attenuation = 1.f;
                if (light.lightType == 0) {
                    dirToLight = normalize(light.loc - hit.hitLoc);
                    float distToLight = length(light.loc - hit.hitLoc);
                    Hit shadowHit;
                    st->rays_shadow++;
                    if (lineCast(C, ray.loc, dirToLight, &shadowHit, st)) {
                        if (distance(shadowHit.hitLoc, hit.hitLoc) < distToLight) continue;
                    }
                    float constant = 1.f, quadratic = 1.f, linear = 1.f;
                    attenuation = light.strength / (constant + linear * distToLight + quadratic * distToLight * distToLight);
                } else if (light.lightType == 1) {
                    dirToLight = -light.dir;
                    attenuation = 1.f;
                    Hit shadowHit;
                    st->rays_shadow++;
                    if (lineCast(C, ray.loc, dirToLight, &shadowHit, st)) continue;
                }
                float NdotL = gmax(dot(normal, dirToLight), 0.0f);
                if (NdotL <= 0.0f) continue;
                vec3 H = normalize(viewDir + dirToLight);
                float NdotH = gmax(dot(normal, H), 0.0f);
                float NdotV = gmax(dot(normal, viewDir), 0.0f);
                float cosTheta = gclamp(dot(normal, -viewDir), 0.0f, 1.0f);
                vec3 F = F0 + (1.0f - F0) * pow5(C, 1.0f - cosTheta);
                float a2 = alpha * alpha;
                float denom = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;
                float D = a2 / (PI * denom * denom);
                float k = (alpha + 1) * (alpha + 1) / 8.0f;
                float G_V = NdotV / (NdotV * (1.0f - k) + k);
                float G_L = NdotL / (NdotL * (1.0f - k) + k);
                float G = G_V * G_L;
                vec3 specular = (D * F * G) / (4.0f * NdotV * NdotL + 0.001f);
                vec3 diffuse = diffuseColor / PI;
                vec3 radiance = light.color * attenuation;
                color = color + (diffuse + specular) * radiance * NdotL * ray.strength;
            }
            if (material.transmissionFactor == 0.f)
                ray.strength = ray.strength * V(gclamp(baseColor.x, 0.f, 1.f), gclamp(baseColor.y, 0.f, 1.f), gclamp(baseColor.z, 0.f, 1.f));
            if (length(ray.strength) < rayStrengthCutoff) ray.isActive = false;
        }
    }
    color_out[0] = color.x; color_out[1] = color.y; color_out[2] = color.z; color_out[3] = 1.f;
    if (P.write_aovs) {
        albedo_out[0] = albedoData.x; albedo_out[1] = albedoData.y; albedo_out[2] = albedoData.z; albedo_out[3] = 1.f;
        normal_out[0] = firstHitNormalData.x / 2.f + 0.5f; normal_out[1] = firstHitNormalData.y / 2.f + 0.5f;
        normal_out[2] = firstHitNormalData.z / 2.f + 0.5f; normal_out[3] = 1.f;
        float d = firstHitDepthData / 255.f;
        depth_out[0] = d; depth_out[1] = d; depth_out[2] = d; depth_out[3] = 1.f;
    }
}

Ctx makeCtx(const RtRefScene* S, int math_mode) {
    Ctx C;
    C.vertices = (const Vertex*)S->vertices; C.meshProj = (const mat4*)S->matrices;
    C.materialList = (const MaterialData*)S->materials; C.boundingBoxList = (const BoundingBox*)S->nodes;
    C.triangleList = (const Triangle*)S->triangles; C.lights = (const Light*)S->lights;
    C.textures = S->textures; C.n_textures = S->n_textures; C.math_mode = math_mode;
    return C;
}

}  // namespace

extern "C" {

// tan(radians(fov) / 2.0) — shader.comp:321, evaluated once on the host.
float rt_ref_tan_half_fov(float fov_deg) { return tanf((fov_deg * 0.017453292519943295f) / 2.0f); }

// Renders region [x0,x1)x[y0,y1), samples [spp_begin, spp_end).  Images are W*H*4 floats, row 0 = bottom.
// prim_tri (W*H int32, may be NULL): triangle id of the sample-0 primary hit (-1 = miss, untouched when 0 is
// outside the sample range).  prim_hit (W*H*3 floats, may be NULL): that hit's hitLoc.
int rt_ref_render(const RtRefScene* S, const RtRefParams* P, float* color, float* albedo, float* normal,
                  float* depth, int32_t* prim_tri, float* prim_hit, RtRefStats* stats) {
    Ctx C = makeCtx(S, P->math_mode);
    const float tanHalf = rt_ref_tan_half_fov(P->fov);
    RtRefStats total; std::memset(&total, 0, sizeof total);
#pragma omp parallel
    {
        RtRefStats st; std::memset(&st, 0, sizeof st);
#pragma omp for schedule(dynamic, 1) collapse(1)
        for (int y = P->y0; y < P->y1; y++) {
            for (int x = P->x0; x < P->x1; x++) {
                size_t idx = (size_t)y * P->width + x;
                shadePixel(C, *P, tanHalf, x, y, color + idx * 4, albedo ? albedo + idx * 4 : nullptr,
                           normal ? normal + idx * 4 : nullptr, depth ? depth + idx * 4 : nullptr,
                           prim_tri ? prim_tri + idx : nullptr, prim_hit ? prim_hit + idx * 3 : nullptr, &st);
            }
        }
#pragma omp critical
        {
            total.rays_primary += st.rays_primary; total.rays_bounce += st.rays_bounce; total.rays_shadow += st.rays_shadow;
            total.node_visits += st.node_visits; total.tri_tests += st.tri_tests;
        }
    }
    if (stats) *stats = total;
    return 0;
}

// Batch lineCast on arbitrary rays (origins/dirs: n*3 floats).  tri_out: triangle id or -1; hit_out: n*3 hitLoc.
int rt_ref_linecast(const RtRefScene* S, const float* origins, const float* dirs, uint64_t n,
                    int32_t* tri_out, float* hit_out, RtRefStats* stats) {
    Ctx C = makeCtx(S, 0);
    RtRefStats total; std::memset(&total, 0, sizeof total);
#pragma omp parallel
    {
        RtRefStats st; std::memset(&st, 0, sizeof st);
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            Hit hit; hit.triIndex = -1; hit.hitLoc = V(0, 0, 0);
            bool h = lineCast(C, V(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]),
                              V(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), &hit, &st);
            tri_out[i] = h ? hit.triIndex : -1;
            if (hit_out) { hit_out[3 * i] = hit.hitLoc.x; hit_out[3 * i + 1] = hit.hitLoc.y; hit_out[3 * i + 2] = hit.hitLoc.z; }
        }
#pragma omp critical
        { total.node_visits += st.node_visits; total.tri_tests += st.tri_tests; }
    }
    if (stats) *stats = total;
    return 0;
}

// Unit-vector entry points for known-answer tests.
float rt_ref_hash13(float x, float y, float z) { return hash13(V(x, y, z)); }
void rt_ref_sincos(float phi, float* s, float* c) { portable_sincos(phi, s, c); }
void rt_ref_refract(const float* I, const float* N, float eta, float* out) {
    vec3 r = refract(V(I[0], I[1], I[2]), V(N[0], N[1], N[2]), eta);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void rt_ref_sample_ggx(const float* N, float r1, float r2, float alpha, int math_mode, float* out) {
    Ctx C; std::memset(&C, 0, sizeof C); C.math_mode = math_mode;
    vec3 h = sampleGGX(C, V(N[0], N[1], N[2]), r1, r2, alpha);
    out[0] = h.x; out[1] = h.y; out[2] = h.z;
}
int rt_ref_intersect_aabb(const float* o, const float* d, const float* mn, const float* mx) {
    return intersectAABB(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), V(mn[0], mn[1], mn[2]), V(mx[0], mx[1], mx[2])) ? 1 : 0;
}
int rt_ref_intersect_triangle(const float* o, const float* d, const float* v0, const float* v1, const float* v2, float* hit) {
    vec3 h = V(0, 0, 0);
    bool r = RayIntersectsTriangle(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), V(v0[0], v0[1], v0[2]), V(v1[0], v1[1], v1[2]), V(v2[0], v2[1], v2[2]), &h);
    hit[0] = h.x; hit[1] = h.y; hit[2] = h.z;
    return r ? 1 : 0;
}
void rt_ref_texture(const RtRefTexture* T, float u, float v, float* out) {
    Ctx C; std::memset(&C, 0, sizeof C); C.textures = T; C.n_textures = 1;
    vec4 t = texture2D(C, 1, vec2{u, v});
    out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
}

}  // extern "C"
