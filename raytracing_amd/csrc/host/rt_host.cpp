// Implementation of rt_host.hpp + the C ABI of include/rt_host.h.
#include "rt_host.hpp"

#include <sys/stat.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "../../../include/rt_host.h"
#include "bvh_build.h"
#include "../rt_compact.h"      // RawBuf, parallel_ranges

namespace rt_host {

// ---- glm subset ---------------------------------------------------------------------------
mat4 operator*(const mat4& a, const mat4& b) {   // glm: Result[c] = A[0]*B[c][0] + A[1]*B[c][1] + A[2]*B[c][2] + A[3]*B[c][3]
    mat4 r;
    for (int c = 0; c < 4; c++)
        for (int k = 0; k < 4; k++)
            r.m[c][k] = a.m[0][k] * b.m[c][0] + a.m[1][k] * b.m[c][1] + a.m[2][k] * b.m[c][2] + a.m[3][k] * b.m[c][3];
    return r;
}
mat4 translate(const mat4& m, vec3 v) {          // Result[3] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2] + m[3]
    mat4 r = m;
    for (int k = 0; k < 4; k++) r.m[3][k] = m.m[0][k] * v.x + m.m[1][k] * v.y + m.m[2][k] * v.z + m.m[3][k];
    return r;
}
mat4 scale(const mat4& m, vec3 v) {
    mat4 r = m;
    for (int k = 0; k < 4; k++) { r.m[0][k] = m.m[0][k] * v.x; r.m[1][k] = m.m[1][k] * v.y; r.m[2][k] = m.m[2][k] * v.z; }
    return r;
}
mat4 mat4_cast(float w, float x, float y, float z) {   // glm/gtc/quaternion.inl mat3_cast
    float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z, qwx = w * x, qwy = w * y, qwz = w * z;
    mat4 r = mat4::identity();
    r.m[0][0] = 1.f - 2.f * (qyy + qzz); r.m[0][1] = 2.f * (qxy + qwz); r.m[0][2] = 2.f * (qxz - qwy);
    r.m[1][0] = 2.f * (qxy - qwz); r.m[1][1] = 1.f - 2.f * (qxx + qzz); r.m[1][2] = 2.f * (qyz + qwx);
    r.m[2][0] = 2.f * (qxz + qwy); r.m[2][1] = 2.f * (qyz - qwx); r.m[2][2] = 1.f - 2.f * (qxx + qyy);
    return r;
}

// ---- ComputeShader (computeShader.cpp) -------------------------------------------------------
ComputeShader* ComputeShader::active = nullptr;
rt_ctx* ComputeShader::activeCtx = nullptr;

static void chk(rt_ctx* c, rt_status s, const char* what) {
    if (s != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error(c));
}

ComputeShader::ComputeShader(const char* /*filePath*/, int device) {
    rt_config cfg{device, 0, 1, 0};
    rt_status s = rt_create(&ID, &cfg);
    // the reference prints compile errors and carries on (computeShader.cpp:33-51); without a GPU there is
    // nothing to carry on with, so this throws
    if (s != RT_OK) throw std::runtime_error(std::string("ComputeShader: ") + rt_last_error(nullptr));
}
ComputeShader::~ComputeShader() { Delete(); }
void ComputeShader::Activate() { active = this; activeCtx = ID; }
void ComputeShader::Delete() {
    if (ID) { if (activeCtx == ID) { activeCtx = nullptr; active = nullptr; } if (owns) rt_destroy(ID); ID = nullptr; }
}
void ComputeShader::setMat4(const std::string& key, const mat4& value) { chk(ID, rt_set_mat4(ID, key.c_str(), &value.m[0][0]), "setMat4"); }
void ComputeShader::setVec4(const std::string& key, vec4 v) { float a[4] = {v.x, v.y, v.z, v.w}; chk(ID, rt_set_vec4(ID, key.c_str(), a), "setVec4"); }
void ComputeShader::setVec3(const std::string& key, vec3 v) { float a[3] = {v.x, v.y, v.z}; chk(ID, rt_set_vec3(ID, key.c_str(), a), "setVec3"); }
void ComputeShader::setVec2(const std::string& key, vec2 v) { float a[2] = {v.x, v.y}; chk(ID, rt_set_vec2(ID, key.c_str(), a), "setVec2"); }
void ComputeShader::setInt(const std::string& key, int value) { if (!ID) return; chk(ID, rt_set_int(ID, key.c_str(), value), "setInt"); }
void ComputeShader::setFloat(const std::string& key, float value) { chk(ID, rt_set_float(ID, key.c_str(), value), "setFloat"); }

// ---- SSBO (SSBO.cpp:5-17) ---------------------------------------------------------------------
std::vector<uint8_t> SSBO::last[6];
// Dry mode (ctx == NULL in rth_load_*): nothing is uploaded, buffers are only kept for rth_scene_buffer and
// texture handles are numbered 1, 2, ... in creation order (what rt_create_texture would return).
static bool g_dry = false;
static uint64_t g_dryHandles = 0;

void SSBO::Bind(const void* data, uint32_t size, uint32_t slot) {
    if (slot < 6) last[slot].assign((const uint8_t*)data, (const uint8_t*)data + size);
    if (g_dry) return;
    if (!ComputeShader::activeCtx) throw std::runtime_error("SSBO::Bind: no active ComputeShader");
    chk(ComputeShader::activeCtx, rt_upload_buffer(ComputeShader::activeCtx, slot, data, size), "SSBO::Bind");
}

// ---- Light (light.cpp) -------------------------------------------------------------------------
std::vector<Light> Light::instances;
Light::Light(vec3 loc, vec3 dir, int lightType, float strength) {
    this->loc = loc; this->dir = dir; this->lightType = lightType; this->strength = strength;
    instances.push_back(*this);
}
std::vector<GPULight> Light::ToGPU(bool& hasSun) {
    hasSun = false;
    std::vector<GPULight> out;
    out.reserve(instances.size());
    for (Light& l : instances) {
        if (l.lightType == LightType::Sun) hasSun = true;
        GPULight g; std::memset(&g, 0, sizeof g);
        vec3 d = normalize(l.dir);      // NaN for point lights (dir = 0), as in the reference
        g.loc = {l.loc.x, l.loc.y, l.loc.z}; g.strength = l.strength;
        g.dir = {d.x, d.y, d.z}; g.lightType = (uint32_t)l.lightType;
        g.color = {l.color.x, l.color.y, l.color.z};
        out.push_back(g);
    }
    return out;
}
void Light::DeleteAll() { instances.clear(); }

// ---- Model (model.cpp) -------------------------------------------------------------------------
std::vector<Model*> Model::instances;
std::vector<ImageRGBA8> Model::textureLog;
double Model::lastFlattenSeconds = 0, Model::lastBuildSeconds = 0, Model::lastUploadSeconds = 0;

Model::Model(const char* f) : file(f) {
    std::vector<uint8_t> text = read_file_bytes(file);
    JSON = Json::parse(std::string(text.begin(), text.end()));
    // getData, model.cpp:149-162
    std::string uri = JSON["buffers"][0]["uri"].string();
    std::string dir = file.substr(0, file.find_last_of('/') + 1);
    data = read_file_bytes(dir + uri);
    traverseNode(0);
    // registered last: a constructor that throws (missing file, bad JSON) must not leave a dangling entry for DeleteAll
    instances.push_back(this);
}

// model.cpp:164-196
std::vector<float> Model::getFloats(const Json& accessor) {
    unsigned int buffViewInd = (unsigned int)accessor.value("bufferView", 1);
    unsigned int count = (unsigned int)accessor["count"].number();
    unsigned int accByteOffset = (unsigned int)accessor.value("byteOffset", 0);
    const std::string& type = accessor["type"].string();
    const Json& bufferView = JSON["bufferViews"][buffViewInd];
    unsigned int byteOffset = (unsigned int)bufferView["byteOffset"].number();
    unsigned int numPerVert;
    if (type == "SCALAR") numPerVert = 1;
    else if (type == "VEC2") numPerVert = 2;
    else if (type == "VEC3") numPerVert = 3;
    else if (type == "VEC4") numPerVert = 4;
    else throw std::invalid_argument("Type is invalid (not SCALAR, VEC2, VEC3, or VEC4)");
    size_t beginningOfData = (size_t)byteOffset + accByteOffset;
    size_t lengthOfData = (size_t)count * 4 * numPerVert;
    if (beginningOfData + lengthOfData > data.size()) throw std::runtime_error("accessor reads past the end of the .bin buffer");
    std::vector<float> out(lengthOfData / 4);
    std::memcpy(out.data(), data.data() + beginningOfData, lengthOfData);
    return out;
}

// model.cpp:198-237
std::vector<uint32_t> Model::getIndices(const Json& accessor) {
    unsigned int buffViewInd = (unsigned int)accessor.value("bufferView", 0);
    unsigned int count = (unsigned int)accessor["count"].number();
    unsigned int accByteOffset = (unsigned int)accessor.value("byteOffset", 0);
    unsigned int componentType = (unsigned int)accessor["componentType"].number();
    const Json& bufferView = JSON["bufferViews"][buffViewInd];
    unsigned int byteOffset = (unsigned int)bufferView["byteOffset"].number();
    size_t beg = (size_t)byteOffset + accByteOffset;
    size_t width = componentType == 5125 ? 4 : (componentType == 5123 || componentType == 5122) ? 2 : 0;
    std::vector<uint32_t> out;
    if (!width) return out;
    if (beg + (size_t)count * width > data.size()) throw std::runtime_error("index accessor reads past the end of the .bin buffer");
    out.resize(count);
    for (unsigned int i = 0; i < count; i++) {
        const unsigned char* p = data.data() + beg + (size_t)i * width;
        if (componentType == 5125) { uint32_t v; std::memcpy(&v, p, 4); out[i] = v; }
        else if (componentType == 5123) { uint16_t v; std::memcpy(&v, p, 2); out[i] = v; }
        else { int16_t v; std::memcpy(&v, p, 2); out[i] = (uint32_t)v; }
    }
    return out;
}

// model.cpp:239-322.  Texture objects are created through rt_create_texture on the active context and the
// returned handle goes straight into MaterialData (the reference stores the GL name here and swaps in the
// bindless handle in Model::BVH, model.cpp:546-559).
void Model::getTextures() {
    materialData.clear();
    std::string dir = file.substr(0, file.find_last_of('/') + 1);
    if (!JSON.has("materials")) return;
    const Json& materials = JSON["materials"];
    for (size_t mi = 0; mi < materials.size(); mi++) {
        const Json& material = materials[mi];
        static const Json emptyObj = Json::parse("{}");
        const Json& pbr = material.has("pbrMetallicRoughness") ? material["pbrMetallicRoughness"] : emptyObj;
        MaterialData md; std::memset(&md, 0, sizeof md);
        for (int i = 0; i < 4; i++) md.baseColorFactor[i] = 1.f;
        if (pbr.has("baseColorFactor")) for (int i = 0; i < 4; i++) md.baseColorFactor[i] = (float)pbr["baseColorFactor"][i].number();
        int baseColorIndex = -1, normalIndex = -1, metallicRoughnessIndex = -1;
        if (pbr.has("baseColorTexture")) baseColorIndex = (int)pbr["baseColorTexture"].value("index", -1);
        if (material.has("normalTexture")) normalIndex = (int)material["normalTexture"].value("index", -1);
        if (pbr.has("metallicRoughnessTexture")) metallicRoughnessIndex = (int)pbr["metallicRoughnessTexture"].value("index", -1);
        md.metallicFactor = (float)pbr.value("metallicFactor", 1.0);
        md.roughnessFactor = (float)pbr.value("roughnessFactor", 1.0);
        md.transmissionFactor = 0; md.ior = -1;
        if (material.has("extensions")) {
            const Json& ext = material["extensions"];
            if (ext.has("KHR_materials_transmission"))   // value("transmissionFactor", 0): int default, int result (Q14)
                md.transmissionFactor = (float)(int)ext["KHR_materials_transmission"].value("transmissionFactor", 0);
            if (ext.has("KHR_materials_ior")) md.ior = (float)ext["KHR_materials_ior"].value("ior", 1.0);
        }
        const int indexList[3] = {normalIndex, baseColorIndex, metallicRoughnessIndex};
        for (int i = 0; i < 3; i++) {
            if (indexList[i] == -1) continue;
            std::string texPath = JSON["images"][(size_t)indexList[i]]["uri"].string();   // texture index used as image index (Q15)
            ImageRGBA8 im = decode_image(read_file_bytes(dir + texPath));      // stbi_load: PNG or JPEG by content
            uint64_t handle = 0;
            if (g_dry) handle = ++g_dryHandles;
            else {
                if (!ComputeShader::activeCtx) throw std::runtime_error("Model: no active ComputeShader for texture upload");
                chk(ComputeShader::activeCtx, rt_create_texture(ComputeShader::activeCtx, im.rgba.data(), im.width, im.height, &handle), "rt_create_texture");
                if (handle != textureLog.size() + 1) throw std::runtime_error("Model: texture handles are not consecutive (textures were created on this context outside the loader)");
                textureLog.push_back(im);
            }
            if (i == 0) { md.normalTexture = handle; md.hasNormalTexture = 1; }
            else if (i == 1) { md.baseColorTexture = handle; md.hasBaseTexture = 1; }
            else { md.metallicRoughnessTexture = handle; md.hasMetallicRoughnessTexture = 1; }
        }
        materialData.push_back(md);
    }
}

// model.cpp:47-72
void Model::loadMesh(unsigned int indMesh) {
    const Json& prim = JSON["meshes"][indMesh]["primitives"][0];
    const Json& at = prim["attributes"];
    const Json& A = JSON["accessors"];
    std::vector<float> pos = getFloats(A[(size_t)at["POSITION"].number()]);
    std::vector<float> nor = getFloats(A[(size_t)at["NORMAL"].number()]);
    std::vector<float> tex = getFloats(A[(size_t)at["TEXCOORD_0"].number()]);
    std::vector<float> tan = getFloats(A[(size_t)at["TANGENT"].number()]);
    size_t n = pos.size() / 3;
    if (nor.size() / 3 < n || tex.size() / 2 < n || tan.size() / 4 < n) throw std::runtime_error("vertex attribute counts differ");
    Mesh mesh;
    mesh.vertices.resize(n);
    for (size_t i = 0; i < n; i++) {          // assembleVertices, model.cpp:324-346
        Vertex& v = mesh.vertices[i]; std::memset(&v, 0, sizeof v);
        v.position = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
        v.normal = {nor[3 * i], nor[3 * i + 1], nor[3 * i + 2]};
        v.color = {1.f, 1.f, 1.f};
        v.texUV[0] = tex[2 * i]; v.texUV[1] = tex[2 * i + 1];
        for (int k = 0; k < 4; k++) v.tangent[k] = tan[4 * i + k];
    }
    mesh.indices = getIndices(A[(size_t)prim["indices"].number()]);
    getTextures();                             // re-run for every mesh, model.cpp:68
    meshes.push_back(std::move(mesh));
}

// model.cpp:74-147
void Model::traverseNode(unsigned int nextNode, mat4 matrix) {
    const Json& node = JSON["nodes"][nextNode];
    vec3 translation(0.f, 0.f, 0.f);
    if (node.has("translation")) { const Json& t = node["translation"]; translation = vec3((float)t[0].number(), (float)t[1].number(), (float)t[2].number()); }
    float qw = 1.f, qx = 0.f, qy = 0.f, qz = 0.f;
    if (node.has("rotation")) { const Json& r = node["rotation"]; qw = (float)r[3].number(); qx = (float)r[0].number(); qy = (float)r[1].number(); qz = (float)r[2].number(); }
    vec3 sc(1.f, 1.f, 1.f);
    if (node.has("scale")) { const Json& s = node["scale"]; sc = vec3((float)s[0].number(), (float)s[1].number(), (float)s[2].number()); }
    mat4 matNode = mat4::identity();
    if (node.has("matrix")) { const Json& m = node["matrix"]; for (size_t i = 0; i < m.size() && i < 16; i++) matNode.m[i / 4][i % 4] = (float)m[i].number(); }
    mat4 trans = translate(mat4::identity(), translation);
    mat4 rot = mat4_cast(qw, qx, qy, qz);
    mat4 sca = scale(mat4::identity(), sc);
    mat4 matNextNode = matrix * matNode * trans * rot * sca;
    if (node.has("mesh")) {
        matricesMeshes.push_back(matNextNode);
        loadMesh((unsigned int)node["mesh"].number());
    }
    if (node.has("children")) {
        const Json& ch = node["children"];
        for (size_t i = 0; i < ch.size(); i++) traverseNode((unsigned int)ch[i].number(), matNextNode);
    }
    if (nextNode + 1 < JSON["nodes"].size()) traverseNode(nextNode + 1);   // Q12
}

// model.cpp:392-406
std::vector<mat4> Model::getMatricesMeshes() {
    std::vector<mat4> out = matricesMeshes;
    mat4 modelTransform = translate(mat4::identity(), pos) * mat4_cast(1.f, 0.f, 0.f, 0.f) * scale(mat4::identity(), scale_);
    for (mat4& m : out) m = m * modelTransform;
    return out;
}

// SSBO::Bind takes a 32-bit size as the reference's does (SSBO.h:9: GLuint).  The reference would wrap silently above
// 4 GiB (33.5 M triangles) and upload a short buffer; the mirror refuses instead (the C ABI itself takes 64 bits).
static uint32_t ssbo_size(size_t bytes, const char* what) {
    if (bytes > 0xffffffffull)
        throw std::length_error(std::string(what) + " buffer is " + std::to_string(bytes) + " bytes: more than SSBO::Bind's 32-bit size holds (SSBO.h:9)");
    return (uint32_t)bytes;
}

// model.cpp:532-613
std::vector<GPUBoundingBox> Model::BVH() {
    auto t0 = std::chrono::steady_clock::now();
    // (one exactly sized, not zero-filled array, written by several threads for big meshes: 2.5 GB at 20 M triangles)
    rt::RawBuf<Triangle> triangles;
    {
        size_t total = 0;
        for (Model* model : instances) for (Mesh& mesh : model->meshes) total += mesh.indices.size() / 3;
        triangles.alloc(total);
    }
    size_t triCursor = 0;
    std::vector<Vertex> verticesList;
    std::vector<mat4> meshMatrices;
    std::vector<MaterialData> materialList;
    uint32_t indiceOffset = 0, meshOffset = 0, materialOffset = 0;
    for (Model* model : instances) {
        std::vector<MaterialData> materialData = model->getMaterialData();
        materialList.insert(materialList.end(), materialData.begin(), materialData.end());
        std::vector<mat4> mm = model->getMatricesMeshes();
        meshMatrices.insert(meshMatrices.end(), mm.begin(), mm.end());
        for (size_t meshIdx = 0; meshIdx < model->meshes.size(); meshIdx++) {
            Mesh& mesh = model->meshes[meshIdx];
            verticesList.insert(verticesList.end(), mesh.vertices.begin(), mesh.vertices.end());
            // material looked up by traversal position, not glTF mesh id (Q16, model.cpp:574)
            const uint32_t materialIdx = (uint32_t)model->JSON["meshes"][meshIdx]["primitives"][0].value("material", 0) + materialOffset;
            const size_t nT = mesh.indices.size() / 3;
            const uint32_t meshIndex = meshOffset + (uint32_t)meshIdx;
            Triangle* dst = triangles.data() + triCursor;
            std::atomic<bool> bad{false};
            rt::parallel_ranges(nT, 1u << 18, [&](uint64_t lo, uint64_t hi, unsigned) {
                for (size_t k = lo; k < hi; k++) {
                    const uint32_t ia = mesh.indices[3 * k], ib = mesh.indices[3 * k + 1], ic = mesh.indices[3 * k + 2];
                    if (ia >= mesh.vertices.size() || ib >= mesh.vertices.size() || ic >= mesh.vertices.size()) { bad = true; return; }
                    Triangle t; std::memset(&t, 0, sizeof t);
                    t.p1 = mesh.vertices[ia].position; t.p2 = mesh.vertices[ib].position; t.p3 = mesh.vertices[ic].position;
                    // indices pass through float (Q13, model.cpp:575): vec3(i) + vec3(offset) -> uvec3
                    t.indices[0] = (uint32_t)((float)ia + (float)indiceOffset);
                    t.indices[1] = (uint32_t)((float)ib + (float)indiceOffset);
                    t.indices[2] = (uint32_t)((float)ic + (float)indiceOffset);
                    vec3 p1(t.p1.x, t.p1.y, t.p1.z), p2(t.p2.x, t.p2.y, t.p2.z), p3(t.p3.x, t.p3.y, t.p3.z);
                    vec3 c = (p1 + p2 + p3) / 3.f;                                  // model.h:54
                    vec3 mn = vmin(vmin(p1, p2), p3), mx = vmax(vmax(p1, p2), p3);  // model.h:55-56
                    t.centroidLoc = {c.x, c.y, c.z}; t.minBounds = {mn.x, mn.y, mn.z}; t.maxBounds = {mx.x, mx.y, mx.z};
                    t.meshIndex = meshIndex; t.materialIndex = materialIdx;
                    dst[k] = t;
                }
            });
            if (bad) throw std::runtime_error("mesh index out of range");
            triCursor += nT;
            indiceOffset += (uint32_t)mesh.vertices.size();
        }
        meshOffset += (uint32_t)model->meshes.size();
        materialOffset += (uint32_t)materialData.size();
    }
    auto t1 = std::chrono::steady_clock::now();
    std::vector<GPUBoundingBox> BVHList = build_bvh(triangles.data(), triangles.size(), 0, /*allowGpu=*/!g_dry);   // a dry run (no context) stays on the host
    auto t2 = std::chrono::steady_clock::now();
    SSBO::Bind(verticesList.data(), ssbo_size(verticesList.size() * sizeof(Vertex), "vertex"), 0);
    SSBO::Bind(meshMatrices.data(), ssbo_size(meshMatrices.size() * sizeof(mat4), "mesh-matrix"), 1);
    SSBO::Bind(materialList.data(), ssbo_size(materialList.size() * sizeof(MaterialData), "material"), 2);
    SSBO::Bind(BVHList.data(), ssbo_size(BVHList.size() * sizeof(GPUBoundingBox), "BVH-node"), 3);
    SSBO::Bind(triangles.data(), ssbo_size(triangles.size() * sizeof(Triangle), "triangle"), 4);
    auto t3 = std::chrono::steady_clock::now();
    lastFlattenSeconds = std::chrono::duration<double>(t1 - t0).count();
    lastBuildSeconds = std::chrono::duration<double>(t2 - t1).count();
    lastUploadSeconds = std::chrono::duration<double>(t3 - t2).count();
    return BVHList;
}

void Model::DeleteAll() {
    for (Model* m : instances) delete m;
    instances.clear();
}

// ---- Scene (Scene.cpp) -------------------------------------------------------------------------
int Scene::currSceneIndex = 0;
std::string Scene::modelsRoot = "models";

void Scene::UnloadScene() { Model::DeleteAll(); Light::DeleteAll(); Model::textureLog.clear(); }

void Scene::LoadScene(ComputeShader* computeShader, int sceneIndex) {
    if (currSceneIndex == sceneIndex) return;
    UnloadScene();
    computeShader->Activate();   // texture uploads happen while Models load
    auto M = [&](const char* rel) { new Model((modelsRoot + "/" + rel).c_str()); };
    switch (sceneIndex) {
    case 1:
        Light(vec3(2, 5, 0), vec3(0.f), LightType::Point, 50.f);
        Light(vec3(-2, 5, 0), vec3(0.f), LightType::Point, 50.f);
        M("sphere/sphere.gltf"); M("cube/cube.gltf"); M("sphere1/sphere.gltf");
        M("glassSphere/sphere.gltf"); M("walls/walls.gltf"); M("roughSphere/roughSphere.gltf");
        break;
    case 2:
        Light(vec3(0, 6.75254f, 0), vec3(0.f), LightType::Point, 10.f);
        M("scene2/scene2.gltf");
        break;
    case 3:
        Light(vec3(0, 6.75254f, 0), vec3(0.f), LightType::Point, 10.f);
        M("box/box.gltf"); M("roughSphere/roughSphere.gltf");
        break;
    case 4:
        Light(vec3(0, 6.75254f, 0), vec3(1, -1, 1), LightType::Sun, 100.f);
        M("scene4/scene4.gltf");
        break;
    case 5:
        Light(vec3(0, 6.75254f, 0), vec3(1, -1, 1), LightType::Sun, 100.f);
        M("burger/burger.gltf");
        break;
    case 6:
        Light(vec3(0, 6.75254f, 0), vec3(1, -1, 1), LightType::Sun, 100.f);
        M("Dragon/Dragon.gltf");
        break;
    case 7:
        Light(vec3(0.f), vec3(0, -1, -1), LightType::Sun, 10.f);
        M("RayTracing/RayTracing.gltf");
        break;
    }
    currSceneIndex = sceneIndex;
    LoadGPUData(computeShader);
}

void Scene::LoadGPUData(ComputeShader* computeShader) {
    computeShader->Activate();
    std::vector<GPUBoundingBox> BVHList = Model::BVH();
    bool hasSun;
    std::vector<GPULight> lights = Light::ToGPU(hasSun);
    SSBO::Bind(lights.data(), ssbo_size(lights.size() * sizeof(GPULight), "light"), 5);
    computeShader->setInt("numLights", (int)lights.size());
    computeShader->setInt("hasSun", hasSun);
    // Scene.cpp:79-84 creates one GL wireframe Box per BVH node for a debug overlay: dropped (no GL).
}

// main.cpp:108-123
void SetFrameUniforms(ComputeShader* cs, const Camera* camera, vec2 screenSize) {
    cs->Activate();
    cs->setVec3("camPos", camera->Position);
    vec3 camForward = normalize(camera->Orientation);
    vec3 camRight = normalize(cross(camForward, vec3(0.f, 1.f, 0.f)));
    vec3 camUp = cross(camRight, camForward);
    cs->setVec3("camForward", camForward);
    cs->setVec3("camRight", camRight);
    cs->setVec3("camUp", camUp);
    cs->setFloat("aspect", screenSize.x / screenSize.y);
    cs->setFloat("fov", camera->FOVdeg);
}

// ---- Renderer (Renderer.cpp) ---------------------------------------------------------------------
int Renderer::currentFrame = 0;
bool Renderer::renderering = false;
std::string Renderer::frameDir;

void Renderer::Start(int /*width*/, int /*height*/, int samplesPerPixel, int maxBounces, const std::string& outDir) {
    currentFrame = 0;
    renderering = true;
    ::mkdir(outDir.c_str(), 0755);
    frameDir = outDir + "/Scene-" + std::to_string(Scene::getCurrSceneIndex()) + "_spp-" + std::to_string(samplesPerPixel) +
               "_bounces-" + std::to_string(maxBounces);   // Renderer.cpp:30 names the .mp4 this way
    ::mkdir(frameDir.c_str(), 0755);
}

// Renderer.cpp:95-124: RGBA32F -> clamp -> *255 truncate -> vertical flip
std::vector<uint8_t> Renderer::FrameRGBA8(rt_ctx* ctx, uint32_t width, uint32_t height) {
    // Renderer.cpp:95-107 (+ the flip of 121-127): done by the core on the GPU, one byte per channel comes back
    std::vector<uint8_t> bytes((size_t)width * height * 4);
    chk(ctx, rt_read_frame_rgba8(ctx, bytes.data(), bytes.size()), "rt_read_frame_rgba8");   // fails if width x height is not the image size
    return bytes;
}

void Renderer::NextFrame(ComputeShader* cs, uint32_t width, uint32_t height, Camera* camera) {
    if (!renderering) return;
    std::vector<uint8_t> bytes = FrameRGBA8(cs->ID, width, height);
    char name[64];
    std::snprintf(name, sizeof name, "/frame_%04d.ppm", currentFrame);
    FILE* fp = std::fopen((frameDir + name).c_str(), "wb");
    if (!fp) throw std::runtime_error("Could not open output file");   // Renderer.cpp:43
    std::fprintf(fp, "P6\n%u %u\n255\n", width, height);
    std::vector<uint8_t> rgb((size_t)width * height * 3);
    for (size_t i = 0; i < (size_t)width * height; i++) { rgb[i * 3] = bytes[i * 4]; rgb[i * 3 + 1] = bytes[i * 4 + 1]; rgb[i * 3 + 2] = bytes[i * 4 + 2]; }
    std::fwrite(rgb.data(), 1, rgb.size(), fp);
    std::fclose(fp);
    UpdateCamera(camera);
    currentFrame++;
    if (currentFrame >= totalFrames) Finished();
}

void Renderer::Finished() { renderering = false; }

// Renderer.cpp:187-193
void Renderer::UpdateCamera(Camera* camera) {
    float PI = 3.14159265359f;
    float x = currentFrame / 120.f - (PI / 6.f);
    vec3 lookAt(0.f, 3.5f, 0.f);
    camera->Position = vec3(std::sin(x) * 15.f, 3.5f, std::cos(x) * 15.f);
    camera->Orientation = normalize(lookAt - camera->Position);
}

}  // namespace rt_host

// =====================================================================================================
// C ABI (include/rt_host.h)
// =====================================================================================================
using namespace rt_host;

namespace rt_host { void set_last_error(const std::string& m); }
namespace {
thread_local std::string g_err;
template <typename F>
rt_status guarded(F&& f) {
    try { f(); return RT_OK; }
    catch (const std::exception& e) { g_err = e.what(); return RT_ERR_SCENE; }
    catch (...) { g_err = "unknown error"; return RT_ERR_SCENE; }
}
struct BorrowedShader {
    ComputeShader shader;
    ComputeShader* cs;
    explicit BorrowedShader(rt_ctx* ctx) : shader(ctx), cs(&shader) {}
};
}  // namespace

namespace rt_host { void set_last_error(const std::string& m) { g_err = m; } }

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }

rt_status rth_load_scene(rt_ctx* ctx, const char* models_root, int32_t scene_index) {
    if (!models_root) { g_err = "NULL argument"; return RT_ERR_INVALID; }
    return guarded([&] {
        g_dry = (ctx == nullptr); g_dryHandles = 0;
        BorrowedShader b(ctx);
        Scene::modelsRoot = models_root;
        Scene::currSceneIndex = 0;          // always reload for an explicit call
        Scene::LoadScene(b.cs, scene_index);
        if (Model::instances.empty()) throw std::runtime_error("scene index has no models (valid: 1..7)");
    });
}

rt_status rth_load_models(rt_ctx* ctx, const char* const* paths, int32_t n_models, const float* lights, int32_t n_lights) {
    if (!paths || n_models <= 0 || (n_lights > 0 && !lights)) { g_err = "bad argument"; return RT_ERR_INVALID; }
    return guarded([&] {
        g_dry = (ctx == nullptr); g_dryHandles = 0;
        BorrowedShader b(ctx);
        Scene::UnloadScene();
        Scene::currSceneIndex = 0;
        b.cs->Activate();
        for (int i = 0; i < n_lights; i++) {
            const float* l = lights + 8 * i;
            Light(vec3(l[0], l[1], l[2]), vec3(l[3], l[4], l[5]), (int)l[6], l[7]);
        }
        for (int i = 0; i < n_models; i++) new Model(paths[i]);
        Scene::LoadGPUData(b.cs);
    });
}

rt_status rth_scene_buffer(uint32_t slot, const void** data, uint64_t* bytes) {
    if (slot > 5 || !data || !bytes) { g_err = "bad argument"; return RT_ERR_INVALID; }
    *data = SSBO::last[slot].data(); *bytes = SSBO::last[slot].size();
    return RT_OK;
}

rt_status rth_load_timings(double out[3]) {
    if (!out) return RT_ERR_INVALID;
    out[0] = Model::lastFlattenSeconds; out[1] = Model::lastBuildSeconds; out[2] = Model::lastUploadSeconds;
    return RT_OK;
}

rt_status rth_build_bvh(const void* triangles, uint64_t n, void** nodes_out, uint64_t* n_nodes_out) {
    if ((!triangles && n) || !nodes_out || !n_nodes_out) { g_err = "bad argument"; return RT_ERR_INVALID; }
    return guarded([&] {
        std::vector<GPUBoundingBox> nodes = build_bvh((const Triangle*)triangles, (size_t)n);
        void* mem = std::malloc(nodes.size() * sizeof(GPUBoundingBox) + 16);
        if (!mem) throw std::runtime_error("out of memory");
        std::memcpy(mem, nodes.data(), nodes.size() * sizeof(GPUBoundingBox));
        *nodes_out = mem; *n_nodes_out = nodes.size();
    });
}

void rth_free(void* p) { std::free(p); }

rt_status rth_set_camera(rt_ctx* ctx, const float position[3], const float orientation[3], float w, float h, float fov) {
    if (!ctx || !position || !orientation) { g_err = "NULL argument"; return RT_ERR_INVALID; }
    return guarded([&] {
        BorrowedShader b(ctx);
        Camera cam((int)w, (int)h, vec3(position[0], position[1], position[2]));
        cam.Orientation = vec3(orientation[0], orientation[1], orientation[2]);
        cam.FOVdeg = fov;
        vec2 ss; ss.x = w; ss.y = h;
        SetFrameUniforms(b.cs, &cam, ss);
    });
}

void rth_orbit_camera(int32_t frame, float position_out[3], float orientation_out[3]) {
    Camera cam(0, 0, vec3(0.f));
    int saved = Renderer::currentFrame;
    Renderer::currentFrame = frame;
    Renderer::UpdateCamera(&cam);
    Renderer::currentFrame = saved;
    position_out[0] = cam.Position.x; position_out[1] = cam.Position.y; position_out[2] = cam.Position.z;
    orientation_out[0] = cam.Orientation.x; orientation_out[1] = cam.Orientation.y; orientation_out[2] = cam.Orientation.z;
}

rt_status rth_frame_rgba8(rt_ctx* ctx, uint32_t width, uint32_t height, uint8_t* out) {
    if (!ctx || !out) { g_err = "NULL argument"; return RT_ERR_INVALID; }
    return guarded([&] {
        std::vector<uint8_t> b = Renderer::FrameRGBA8(ctx, width, height);
        std::memcpy(out, b.data(), b.size());
    });
}

rt_status rth_decode_png(const char* path, uint8_t** rgba8_out, uint32_t* w, uint32_t* h) { return rth_decode_image(path, rgba8_out, w, h); }

rt_status rth_decode_image(const char* path, uint8_t** rgba8_out, uint32_t* w, uint32_t* h) {
    if (!path || !rgba8_out || !w || !h) { g_err = "NULL argument"; return RT_ERR_INVALID; }
    return guarded([&] {
        ImageRGBA8 im = decode_image(read_file_bytes(path));
        uint8_t* mem = (uint8_t*)std::malloc(im.rgba.size() + 16);
        if (!mem) throw std::runtime_error("out of memory");
        std::memcpy(mem, im.rgba.data(), im.rgba.size());
        *rgba8_out = mem; *w = im.width; *h = im.height;
    });
}

}  // extern "C"
