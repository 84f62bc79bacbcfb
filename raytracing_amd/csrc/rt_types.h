// Device/host shared record layouts of the MI355X path-tracing core.
//
// Two families:
//  * Ref*  — the reference's std430 SSBO records exactly as SSBO::Bind uploads them
//            (/root/reference/Raytracing/VBO.h:9-17, model.h:8-39,41-70,130-142, light.h:14-23).
//            They are the INPUT format of rt_upload_buffer and stay byte-compatible (drop-in).
//  * C*    — the compacted HBM layout the kernels read (DESIGN.md "Data layout in HBM").
#pragma once
#include <stdint.h>

namespace rt {

struct f3 { float x, y, z; };

// ---- reference input records ---------------------------------------------------------------
struct RefVertex {            // 80 B  VBO.h:9-17 <-> shader.comp:14-20
    f3 position; float pad1;
    f3 normal;   float pad2;
    f3 color;    float pad3;
    float texUV[2]; float pad4[2];
    float tangent[4];
};
struct RefMaterial {          // 80 B  model.h:8-39 <-> shader.comp:22-39
    float baseColorFactor[4];
    uint64_t baseColorTexture; int32_t hasBaseTexture; float transmissionFactor;
    uint64_t normalTexture; int32_t hasNormalTexture; float metallicFactor;
    uint64_t metallicRoughnessTexture; int32_t hasMetallicRoughnessTexture; float roughnessFactor;
    float ior; float pad1[3];
};
struct RefNode {              // 48 B  model.h:130-142 <-> shader.comp:41-46
    f3 minLoc; float pad1;
    f3 maxLoc; uint32_t isLeaf;
    int32_t children[4];
};
struct RefTriangle {          // 128 B model.h:41-70 <-> shader.comp:48-59
    uint32_t indices[3]; uint32_t pad0;
    f3 p1; float pad_1;
    f3 p2; float pad_2;
    f3 p3; float pad_3;
    f3 centroidLoc; float pad_4;
    f3 minBounds; float pad_5;
    f3 maxBounds; uint32_t meshIndex;
    uint32_t materialIndex; uint32_t pad1[2]; uint32_t tail;
};
struct RefLight {             // 48 B  light.h:14-23 <-> shader.comp:69-75
    f3 loc; float strength;
    f3 dir; uint32_t lightType;
    f3 color; float pad1;
};
struct RefMat4 { float m[4][4]; };   // [col][row], glm::mat4

static_assert(sizeof(RefVertex) == 80, "std430 Vertex");
static_assert(sizeof(RefMaterial) == 80, "std430 MaterialData");
static_assert(sizeof(RefNode) == 48, "std430 BoundingBox");
static_assert(sizeof(RefTriangle) == 128, "std430 Triangle");
static_assert(sizeof(RefLight) == 48, "std430 Light");
static_assert(sizeof(RefMat4) == 64, "mat4");

// ---- compacted device records --------------------------------------------------------------
// One record per INTERNAL reference node, holding both children's boxes, so a visit is one 64-byte
// fetch (4 x dwordx4) and leaves need no record of their own.
//   child >= 0 : index of an internal CNode
//   child <  0 : leaf, code = -child-1 = first*8 + count   (count 0..4 triangles at CTri[first..])
// Box of an internal child = the reference's own box of that node (bit-identical slab result).
// Box of a leaf child      = world-space bounds of its triangles, padded (used only by traversalMode 0;
//                            the reference never tests a leaf's box: model.h:138-139).
struct CNode {
    float lo0[3]; float hi0[3];
    float lo1[3]; float hi1[3];
    int32_t child0, child1;
    uint32_t pad[2];
};
static_assert(sizeof(CNode) == 64, "CNode");

// Wide node: TWO levels of the reference tree per fetch.  A slot is a grandchild of the binary node the record stands for
// (or a child that is a leaf), with the box that gates it: the reference's own box of that grandchild if it is an internal
// node, the padded world bounds of a leaf clipped to the box of the child above it.  The child's own box needs no test:
// the reference's boxes are nested (a parent's box is the exact union of its children's) and the slab test is monotone
// in the box bounds under IEEE rounding, so a ray that passes a grandchild's box passes the child's.
//   ref >= 0 : index of the slot's WNode;  ref < -1 : leaf code as in CNode;  ref == -1 : empty slot (inverted box)
struct WNode {
    float box[4][6];               // slot k: lo.xyz, hi.xyz
    int32_t ref[4];
    uint32_t pad[4];
};
static_assert(sizeof(WNode) == 128, "WNode");

// Pre-transformed triangle in reference visit order (right-child-first DFS, leaf slots in order), so
// the array index is the reference's visit rank and a leaf is a contiguous run.
struct CTri {
    float v0[3]; float e1[3]; float e2[3];   // v0 = M*p1, e1 = M*p2 - v0, e2 = M*p3 - v0 (same float ops as shader.comp:263-265,105-106)
    int32_t refIndex;                         // index into the reference Triangle[] (slot 4)
    uint32_t pad[2];
};
static_assert(sizeof(CTri) == 48, "CTri");

// Shading inputs per reference triangle (indexed by refIndex): everything shader.comp:342-357 fetches through
// Triangle / Vertex / meshProj, resolved once at upload.  Split by use — k_shade is bound by the number of
// per-lane load instructions, and only textured / normal-mapped materials need the second record.
struct CShade {                    // 80 B = 5 x dwordx4
    float a[3], b[3], c[3];        // Triangle.p1..p3 (mesh-local, used by barycentric())
    float n1[3], n2[3], n3[3];     // vec3(meshProj * vec4(normal, 0))
    uint32_t materialIndex;
    uint32_t pad;
};
static_assert(sizeof(CShade) == 80, "CShade");
struct CShadeUV {                  // 48 B, fetched only when the material samples a texture
    float uv1[2], uv2[2], uv3[2];
    uint32_t i1, i2, i3;           // vertex ids (tangent fetch for normal-mapped materials)
    uint32_t pad[3];
};
static_assert(sizeof(CShadeUV) == 48, "CShadeUV");

// The scalars of MaterialData the integrator reads on every hit (model.h:8-39), 2 x dwordx4 instead of 5; texture
// handles stay in the reference record (slot 2), read only when a flag says so.
struct CMat {
    float baseColor[3]; float transmission;
    float metallic, roughness, ior;
    uint32_t flags;                // bit0 hasBaseTexture == 1, bit1 hasNormalTexture == 1, bit2 hasMetallicRoughnessTexture == 1
};
static_assert(sizeof(CMat) == 32, "CMat");

struct TexDesc { const uint8_t* rgba8; uint32_t width, height; };

#define RT_MAX_LIGHTS 8
#define RT_MAX_STREAMS 8
#define RT_LDS_STACK 8           // per-lane stack entries kept in LDS
#define RT_SPILL_STACK 56        // further entries in scratch; total = 64 = shader.comp:246

// Scene + frame constants handed to every kernel by value (kernarg segment -> SGPRs).
struct SceneView {
    const CNode* nodes;
    const CTri* tris;
    const CShade* shade;
    const CShadeUV* shadeUV;
    const CMat* cmat;
    const RefMaterial* materials;
    const RefVertex* vertices;
    const RefLight* lights;
    const TexDesc* textures;
    uint32_t n_textures;
    int32_t rootChild;             // >= 0 internal CNode index, < 0 leaf code (root is a leaf)
    float rootLo[3], rootHi[3];    // reference box of node 0 (tested first, shader.comp:249-256)
    float cullAbs;                 // absolute slack of the distance cull (traversalMode 0)
    int32_t exactMode;             // 1 = reference-order exhaustive traversal
    const WNode* wnodes;           // two-level nodes of the same tree (null: not derived for this scene); root = wnodes[0]
};

struct FrameParams {
    int32_t width, height;         // image size
    int32_t rx, ry, rw, rh;        // region
    int32_t tilesX;                // region width in 8x8 tiles
    int32_t numLights, hasSun;
    int32_t anyPoint;              // a point light is among the numLights lights (k_shadow then needs the shade location)
    RefLight lights[RT_MAX_LIGHTS];  // slot 5, by value: kernels that index it uniformly read it with scalar loads
    float camPos[3], camForward[3], camRight[3], camUp[3];
    float aspect, tanHalfFov;
    int32_t samplesPerPixel, maxBounces;
    int32_t gridN;                 // int(ceil(sqrt(float(spp)))), shader.comp:309-310
    float initialStrength, strengthCutoff;   // shader.comp:306-307
};

struct DeviceStats {
    unsigned long long rays_primary, rays_bounce, rays_shadow, rays_shadow_skipped;
    unsigned long long node_visits, tri_tests;                 // closest-hit (k_extend) traversal
    unsigned long long shadow_node_visits, shadow_tri_tests;   // shadow (k_shadow) traversal
    unsigned long long max_ray_steps;                          // most node visits + triangle tests by one closest-hit ray
    unsigned long long bounce_node_visits, bounce_tri_tests;   // closest-hit traversal of bounces >= 1 only
    unsigned long long wave_node_steps, wave_tri_steps;        // wave-level executions of the node / triangle step there
    unsigned long long dbg2[4];
    unsigned long long dbg[8];                                 // instrumented refill kernel: scheduling census (RT_DEBUG=1 prints it)
    unsigned long long late_overflow;                          // k_shade: a bounce's queue was longer than the compact part of the log holds
};

}  // namespace rt
