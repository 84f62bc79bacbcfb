// GL-free rebuild of the reference's host-side callers of the integrator, keeping their class names
// and call shapes (Scene / Model / Light / Camera / ComputeShader / SSBO / Renderer), over the C ABI of
// include/rt_core.h.  Reference files: /root/reference/Raytracing/{Scene,model,light,camera,SSBO,
// computeShader,Renderer}.{h,cpp}.  glm is replaced by the few float32 operations those files use,
// written with glm's own formulas (normalize = v * inversesqrt(dot(v,v)), column-major mat4).
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/rt_core.h"
#include "../rt_types.h"
#include "json.h"
#include "jpeg.h"     // + png.h

namespace rt_host {

struct vec2 { float x = 0, y = 0; };
struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit vec3(float a) : x(a), y(a), z(a) {}
};
struct vec4 { float x = 0, y = 0, z = 0, w = 0; };
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 x, vec3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }   // glm: v * inversesqrt(dot(v, v))
inline vec3 vmin(vec3 a, vec3 b) { return {b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z}; }
inline vec3 vmax(vec3 a, vec3 b) { return {a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z}; }

struct mat4 {   // column-major, m[col][row]
    float m[4][4];
    static mat4 identity() { mat4 r; for (int c = 0; c < 4; c++) for (int k = 0; k < 4; k++) r.m[c][k] = c == k ? 1.f : 0.f; return r; }
};
mat4 operator*(const mat4& a, const mat4& b);
mat4 translate(const mat4& m, vec3 v);
mat4 scale(const mat4& m, vec3 v);
mat4 mat4_cast(float w, float x, float y, float z);

using Vertex = rt::RefVertex;            // VBO.h:9-17
using MaterialData = rt::RefMaterial;    // model.h:8-39
using Triangle = rt::RefTriangle;        // model.h:41-70
using GPUBoundingBox = rt::RefNode;      // model.h:130-142
using GPULight = rt::RefLight;           // light.h:14-23

// computeShader.h:11-31 — the "program" is the HIP core; the file path is accepted and ignored.
class ComputeShader {
public:
    explicit ComputeShader(const char* filePath, int device = 0);
    explicit ComputeShader(rt_ctx* borrowed) : ID(borrowed), owns(false) {}   // wraps a caller-owned context
    ~ComputeShader();
    rt_ctx* ID = nullptr;
    bool owns = true;
    void Activate();                       // glUseProgram: makes this the target of SSBO::Bind
    void Delete();
    void setMat4(const std::string& key, const mat4& value);
    void setVec4(const std::string& key, vec4 value);
    void setVec3(const std::string& key, vec3 value);
    void setVec2(const std::string& key, vec2 value);
    void setInt(const std::string& key, int value);
    void setFloat(const std::string& key, float value);
    static ComputeShader* active;
    static rt_ctx* activeCtx;              // set by Activate() or by the C entry points
};

// SSBO.h:7-10
class SSBO {
public:
    static void Bind(const void* data, uint32_t size, uint32_t slot);
    // host copies of what was bound last (for rth_scene_buffer)
    static std::vector<uint8_t> last[6];
};

struct LightType { static constexpr int Point = 0; static constexpr int Sun = 1; };   // light.h:7-12

// light.h:25-40, light.cpp
class Light {
public:
    Light(vec3 loc, vec3 dir, int lightType, float strength = 1.f);
    static std::vector<GPULight> ToGPU(bool& hasSun);
    static void DeleteAll();
    static std::vector<Light> instances;
private:
    vec3 loc; vec3 dir = vec3(0.f); vec3 color = vec3(1.f); float strength = 1.f; int lightType = 0;
};

struct Mesh { std::vector<Vertex> vertices; std::vector<uint32_t> indices; };   // mesh.h (GL objects dropped)

// model.h:144-221, model.cpp
class Model {
public:
    explicit Model(const char* file);
    std::vector<Mesh>& getMeshes() { return meshes; }
    std::vector<mat4> getMatricesMeshes();
    std::vector<MaterialData> getMaterialData() { return materialData; }
    static std::vector<GPUBoundingBox> BVH();
    static void DeleteAll();
    static std::vector<Model*> instances;
    // every texture handed to rt_create_texture since the last UnloadScene, in creation order (handle k + 1 = entry k): what
    // MultiGPU::ReplicateScene re-creates on the other devices so that the material records' handles mean the same there
    static std::vector<ImageRGBA8> textureLog;
    static double lastFlattenSeconds, lastBuildSeconds, lastUploadSeconds;
private:
    vec3 pos = vec3(0.f), scale_ = vec3(1.f);
    std::vector<unsigned char> data;
    Json JSON;
    std::string file;
    std::vector<Mesh> meshes;
    std::vector<mat4> matricesMeshes;
    std::vector<MaterialData> materialData;
    void loadMesh(unsigned int indMesh);
    void traverseNode(unsigned int nextNode, mat4 matrix = mat4::identity());
    std::vector<float> getFloats(const Json& accessor);
    std::vector<uint32_t> getIndices(const Json& accessor);
    void getTextures();
};

// camera.h:17-50 (inputs / GL matrices dropped; only what reaches the kernel)
class Camera {
public:
    vec3 Position;
    vec3 Orientation = vec3(0.0f, 0.0f, -1.0f);
    vec3 Up = vec3(0.0f, 1.0f, 0.0f);
    float FOVdeg = 45.f;
    int width, height;
    Camera(int width, int height, vec3 position) : Position(position), width(width), height(height) {}
};

// Scene.h:7-20
class Scene {
public:
    static void LoadScene(ComputeShader* computeShader, int sceneIndex);
    static int getCurrSceneIndex() { return currSceneIndex; }
    static void UnloadScene();
    static std::string modelsRoot;         // the reference's working-directory-relative "models/"
    static void LoadGPUData(ComputeShader* computeShader);
    static int currSceneIndex;
};

// main.cpp:108-123 — per-frame uniforms
void SetFrameUniforms(ComputeShader* cs, const Camera* camera, vec2 screenSize);

// Renderer.h:19-48.  FFmpeg is not in this image: frames are written as binary PPM files
// "<dir>/Scene-<i>_spp-<s>_bounces-<b>/frame_%04d.ppm" with the reference's 8-bit conversion.
class Renderer {
public:
    static void Start(int width, int height, int samplesPerPixel, int maxBounces, const std::string& outDir = "Renders");
    static void NextFrame(ComputeShader* cs, uint32_t width, uint32_t height, Camera* camera);
    static bool isRenderering() { return renderering; }
    static void UpdateCamera(Camera* camera);
    static void Finished();
    static std::vector<uint8_t> FrameRGBA8(rt_ctx* ctx, uint32_t width, uint32_t height);
    static int currentFrame;
    static constexpr int fps = 60;
    static constexpr int totalFrames = 120;
private:
    static bool renderering;
    static std::string frameDir;
};

// Multi-GPU frame driver (no reference counterpart; include/rt_host.h, rth_multi_*).  One context per entry of
// `devices`, scene replicated, one host thread per context per frame, one RCCL reduce of the colour image.
struct Shard { uint32_t x = 0, y = 0, w = 0, h = 0, r0 = 0, r1 = 0; bool empty() const { return w == 0 || h == 0 || r1 <= r0; } };
// `bandRanks` row bands (boundaries on multiples of 8 rows) x world / bandRanks contiguous sample ranges; == distributed.partition
Shard partition(uint32_t width, uint32_t height, uint32_t spp, int rank, int world, int bandRanks);

class MultiGPU {
public:
    explicit MultiGPU(const std::vector<int>& devices);
    ~MultiGPU();
    MultiGPU(const MultiGPU&) = delete;
    MultiGPU& operator=(const MultiGPU&) = delete;
    size_t size() const { return shaders.size(); }
    ComputeShader* shader(size_t i) { return shaders[i]; }
    void ReplicateScene();                                  // after Scene::LoadScene / LoadGPUData on shader(0)
    void setInt(const std::string& key, int value);
    void resize(uint32_t w, uint32_t h);
    void SetFrameUniforms(const Camera* camera, vec2 screenSize);
    void RenderFrame(uint32_t spp, int bandRanks = 1, bool mergeAovs = false);   // merged frame on shader(0)
    bool usedRccl() const { return lastUsedRccl; }
private:
    struct Impl;
    Impl* impl = nullptr;
    std::vector<ComputeShader*> shaders;
    std::vector<int> devices;
    uint32_t width = 0, height = 0;
    int framesRendered = 0;
    bool kernelsAgreed = false, lastUsedRccl = false;
};

}  // namespace rt_host
