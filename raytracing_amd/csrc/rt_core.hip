// librt_core.so — C ABI of the MI355X path-tracing core (include/rt_core.h).
// Host orchestration of the wavefront kernels in rt_kernels.hip.h.  gfx950 only; no CPU fallback.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rt_core.h"
#include "rt_compact.h"
#include "rt_kernels.hip.h"
#include "rt_bvh_sweep.hip.h"

using namespace rt;

namespace {

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

struct Texture { DevBuf data; uint32_t w = 0, h = 0; };

// ---- rt_trace_rays_noalloc: device memory that is part of the library image, not of the heap -----------------------
// Everything rt_trace_rays keeps in hipMalloc'ed buffers (ray queue, hit records, queue counter, traversal-stack
// overflow, staged inputs and outputs) lives in this file-scope array for the allocation-free variant.  Each GPU the
// library runs on holds its own copy until the process ends.  Layout for n rays (offsets in bytes, all 256-aligned):
//   [0, 512)  queue counter + DeviceStats | rays 32 n | hits 8 n | origins 12 n | dirs 12 n | tri ids 4 n | hit locs 12 n |
//   stack overflow 8 x RT_SPILL_STACK x n (the deepest tree the shader's 64-entry stack can hold)
constexpr uint64_t kNoAllocMaxRays = RT_NOALLOC_MAX_RAYS;
constexpr uint64_t kScratchHeader = 512;
constexpr uint64_t kScratchBytes = kScratchHeader + kNoAllocMaxRays * (32 + 8 + 12 + 12 + 4 + 12 + 8ull * RT_SPILL_STACK) + 7 * 256;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
// `scratch` is a host-side stand-in in host code, not a device address: the address of each device's copy is looked up
// once per device (rt_create, with that device current) and only these pointers are ever handed to kernels and copies.
constexpr int kMaxDevices = 64;
unsigned char* g_scratchPtr[kMaxDevices] = {};
std::once_flag g_scratchOnce[kMaxDevices];      // two threads creating contexts on one device look the address up once
hipError_t g_scratchErr[kMaxDevices] = {};

}  // namespace

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // host copies of the reference buffers (slot uploads copy immediately; compaction is lazy)
    // The six uploaded buffers as the caller handed them over (rt_upload_buffer copies at once, as glBufferData does).
    // Copied by several threads when large: the first touch of gigabytes of fresh pages by one thread cost 1.5 s at config 4.
    struct HostBuf {
        uint8_t* p = nullptr; size_t n = 0, cap = 0;
        HostBuf() = default;
        HostBuf(const HostBuf&) = delete;
        HostBuf& operator=(const HostBuf&) = delete;
        ~HostBuf() { std::free(p); }
        const uint8_t* data() const { return p; }
        size_t size() const { return n; }
        void assign(const uint8_t* src, size_t bytes) {
            if (bytes > cap || bytes < cap / 2) {
                std::free(p); p = nullptr; cap = 0;
                if (bytes) {
                    // large buffers: 2 MB-aligned and advised as huge pages, so that the first touch is a few thousand
                    // page faults instead of a million
                    const size_t big = (size_t)64 << 20, align = bytes >= big ? ((size_t)2 << 20) : 64;
                    p = (uint8_t*)std::aligned_alloc(align, (bytes + align - 1) / align * align);
                    if (!p) throw std::bad_alloc();
                    if (bytes >= big) (void)madvise(p, (bytes + align - 1) / align * align, MADV_HUGEPAGE);
                    cap = bytes;
                }
            }
            n = bytes;
            uint8_t* dst = p;
            rt::parallel_ranges(bytes, (uint64_t)64 << 20, [&](uint64_t lo, uint64_t hi, unsigned) { if (hi > lo) std::memcpy(dst + lo, src + lo, hi - lo); });
        }
    };
    HostBuf slotHost[6];
    bool slotSet[6] = {false, false, false, false, false, false};
    bool sceneDirty = true, lightsDirty = true, materialsDirty = true, texturesDirty = true;

    // uniforms (shader.comp:85-102)
    int32_t numLights = 0, hasSun = 0, samplesPerPixel = 1, maxBounces = 1;
    float camPos[3] = {0, 0, 0}, camForward[3] = {0, 0, -1}, camRight[3] = {1, 0, 0}, camUp[3] = {0, 1, 0};
    float aspect = 1.f, fov = 45.f;
    float offset[2] = {0, 0};
    int32_t traversalMode = 0, timeKernels = 0, countVisits = 0, samplesPerPass = 64, blocksPerCU = 64, sampleGroup = -1, pathMemPercent = 75, logDenseBounces = 0, lateLogPercent = 25;
    uint32_t lateLogFallbacks = 0;
    bool lateLogOff = false;      // a frame overflowed the compact part of the log: this context keeps a dense log from then on

    // device scene
    DevBuf dNodes, dWNodes, dTris, dShade, dShadeUV, dMaterials, dCMat, dVertices, dLights, dTexDescs;
    std::vector<Texture> textures;
    CompactScene cs;
    uint64_t nLights = 0, nMaterials = 0;

    // images
    uint32_t width = 0, height = 0;
    DevBuf img[4];
    DevBuf primTri, primHit;
    DevBuf rgbStage;            // packed RGB staging of rt_read_image_rgb / rt_write_image_rgb
    bool primValid = false;

    // path state
    // Up to kMaxSets sets of per-pass path buffers: consecutive passes run on `passStreams` streams, so that the small, tail-dominated
    // launches of one pass's late bounces overlap the large launches of the next pass (only k_resolve is ordered).
    // The per-slot arrays of ALL sets live in one allocation per array (set k's part starts at k x the set's size): the
    // same memory serves four sets of S samples and two sets of 2 S, so trying and switching the pass shape allocates nothing
    // (hipFree + hipMalloc of ~100 GB cost seconds).  Per set: the queue counters and the traversal-stack overflow area.
    enum { A_RAY0, A_RAY1, A_RAYS0, A_RAYS1, A_HIT, A_SHADOWREC, A_SHADELOC, A_CONTRIB, A_DEPTH, A_LATE, A_LATENEXT, A_LATEHEAD, A_COUNT };
    DevBuf arena[A_COUNT];
    struct PathSet { DevBuf counters, overflow; };
    static constexpr int kMaxSets = 8;
    PathSet set[kMaxSets];
    DevBuf accum, dStats;
    hipStream_t extra[kMaxSets] = {};     // streams of sets 1.. (set 0 uses `stream`)
    hipEvent_t evResolve[kMaxSets] = {}, evInit = nullptr;
    int32_t passStreams = 4, autoPass = 1, fusePairs = 1;
    // dynamic-fetch traversal kernels (rt_trace_refill.hip.h): refill 0 = never, 1 = always, -1 = chosen per scene.
    int32_t refill = -1, refillIdle = 16, refillNodeMin = 28, refillBlocksPerCU = 0;
    // The sun's shadow rays of the first `packetBounces` bounces are traced by the packet kernel (a wave walks the tree once
    // for 64 consecutive rays, scalar loads; rt_trace_packet.hip.h): bounce 0's start at the camera-ray hits of a few pixels.
    // -1: bounce 0 where the compact BVH is at most 1 MB — a packet step is one dependent scalar-load round trip per wave,
    // which pays while the tree sits in every L2 (scene 4: frame -0.7 %, those rays -11 %) and not beyond (scene 7 +-0,
    // the 20 M-triangle field +-0 for bounce 0 and +60 % on the shadow rays with bounce 1 included).
    int32_t packetBounces = -1;
    int32_t refillFromBounce = 1;   // closest-hit rays of bounces below this use the batch kernel even where the refill kernel was chosen (camera rays always do)
    int32_t refillTriMin = 0, refillNodeLow = 0, refillDefer = 0;   // 0 = the kernels' defaults (rt_trace_refill.hip.h, RefillSched)   // refillBlocksPerCU 0 = automatic (refill_blocks)
    // refill = -1: which traversal kernels are faster depends on the scene, and not the same way for closest-hit and
    // shadow rays (closest hit: the refill kernel on every scene measured; shadow rays: the one-sun batch kernel and the
    // refill kernel within a few per cent of each other on scenes 3, 4 and 7, the refill kernel by far on the
    // 20 M-triangle field), so the first four sizeable frames after a scene upload are rendered with the four
    // combinations and the fastest (rays per millisecond) is kept.  Images do not depend on the choice.
    uint64_t allocations = 0;   // hipMalloc calls so far (a frame that allocated is not a timing sample)
    int tuneState = 0;          // 0..3: the next sizeable frame samples combination kTuneTrial[tuneState]; 4: decided
    double tuneRate[4] = {0.0, 0.0, 0.0, 0.0};
    uint64_t sceneKey = 0;      // content key of the scene on the device (scene_key): the trial's result is remembered under it
    int32_t tuneCache = 1;      // 0: always run the trial (knob)
    // Pass shape.  Default: `passStreams` (4) streams of up to `samplesPerPass` (64) samples.  Scenes whose launches end in
    // long latency-bound tails (scene 7's glass: a ray inside it walks ~750 dependent steps) render faster with half the
    // streams and twice the samples per pass (C5 -4 %, C3 -4 %), others slower (C2 +3 %, C4 +5 %): measured per scene, after
    // the kernel trial, on one more sizeable frame (plus the frame that re-allocates the path buffers, which is no sample).
    // Not tried when the caller set passStreams or samplesPerPass.
    bool streamsUser = false, sppUser = false;
    int shapeState = 0;         // 0: not measured yet, 1: measuring, 2: decided (shapeChoice)
    int shapeChoice = 0;        // 1: half the streams, twice the samples per pass
    double tuneRateBase = 0.0;  // rays per ms of the kept kernel combination with the default shape
    int tuneChoice = 0;         // bit 0: refill kernel for closest-hit rays, bit 1: for shadow rays
    int32_t kernelChoice = -1;  // >= 0: this combination (bit 0 / bit 1 as tuneChoice) whatever was measured: every rank of a
                                // multi-GPU job is given rank 0's choice, so that shards cost the same everywhere
    int32_t sunKernel = 1;      // 0: the general k_shadow also for one-sun scenes (A/B knob)
    // The refill kernels walk two-level nodes (WNode) derived from the same tree, when it qualifies (derive_wide):
    // 1 always, 0 never, -1 for BVHs far larger than the L2s (where they pay: config 4 -8 %)
    int32_t wideNodes = -1;
    bool wideReady = false;
    bool uvReady = false;       // the CShadeUV table of the scene on the device has been built (it is only when a material is textured)     // derive_wide has run for the scene on the device (its result may be "does not qualify")
    bool anyTextured = true, anyTransmissive = true;   // which k_shade variant the scene needs (set with the materials)
    uint32_t nslotsAlloc = 0, pendLights = 0, countersAlloc = 0;

    struct EvPair { hipEvent_t a = nullptr, b = nullptr; int kind = 0; bool used = false; };
    std::vector<EvPair> evRing;
    rt_stats stats{};
    int numCUs = 256;
    uint64_t deviceBytes = 288ull << 30;
};

namespace {

thread_local std::string g_createErr;

// What the per-scene kernel trial kept, by scene content and device: a scene that comes back (the reference's interactive
// scene switching, main.cpp:308-321; a second context on the same GPU) skips the four trial frames.  Process-wide.
struct TuneCache { std::mutex mu; std::unordered_map<uint64_t, int> choice; };
TuneCache& tune_cache() { static TuneCache t; return t; }

rt_status fail(rt_ctx* c, rt_status s, const std::string& m) { if (c) c->err = m; else g_createErr = m; return s; }

#define HIPCHK(c, call)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) return fail(c, RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

rt_status ensure(rt_ctx* c, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return RT_OK;
    b.release();
    if (bytes == 0) bytes = 16;
    HIPCHK(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    c->allocations++;
    return RT_OK;
}

rt_status upload(rt_ctx* c, DevBuf& b, const void* src, size_t bytes) {
    rt_status s = ensure(c, b, bytes);
    if (s) return s;
    if (bytes) HIPCHK(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    return RT_OK;
}

const size_t kRecordSize[6] = {80, 64, 80, 48, 128, 48};

// BVHs this much larger than the L2s (32 MB) are traversed out of the Infinity Cache and HBM: the refill kernels and the
// two-level nodes are their defaults
constexpr size_t kBigBvhBytes = (size_t)64 << 20;
constexpr size_t kSmallBvhBytes = (size_t)1 << 20;
size_t bvh_bytes(const rt_ctx* c);
int packet_bounces(const rt_ctx* c);
size_t bvh_bytes(const rt_ctx* c) { return c->cs.nodes.size() * sizeof(CNode) + (size_t)c->cs.nTris * sizeof(CTri); }
int packet_bounces(const rt_ctx* c) { return c->packetBounces >= 0 ? (int)c->packetBounces : (bvh_bytes(c) <= kSmallBvhBytes ? 1 : 0); }
bool shape_user(const rt_ctx* c) { return c->streamsUser || c->sppUser; }      // the caller chose the pass shape: no trial of it
bool want_wide(const rt_ctx* c) { return c->wideNodes > 0 || (c->wideNodes < 0 && bvh_bytes(c) > kBigBvhBytes); }

// FNV-1a over the sizes of slots 0..4 and up to 256 evenly spaced 64-byte samples of the node and triangle buffers
uint64_t scene_key(const rt_ctx* c) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
    const int32_t dev = c->device;
    mix(&dev, sizeof dev);
    for (int s = 0; s < 5; s++) { const uint64_t n = c->slotHost[s].size(); mix(&n, sizeof n); }
    for (int s : {RT_SLOT_BVH_NODES, RT_SLOT_TRIANGLES}) {
        const auto& B = c->slotHost[s];
        const size_t chunks = B.size() / 64, step = std::max<size_t>(1, chunks / 256);
        for (size_t k = 0; k < chunks; k += step) mix(B.data() + k * 64, 64);
    }
    return h;
}

rt_status sync_scene(rt_ctx* c) {
    for (int s = 0; s < 6; s++)
        if (!c->slotSet[s]) return fail(c, RT_ERR_NOT_READY, "slot " + std::to_string(s) + " has not been uploaded");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->sceneDirty) {
        const auto& V = c->slotHost[RT_SLOT_VERTICES]; const auto& M = c->slotHost[RT_SLOT_MESH_MATRICES];
        const auto& N = c->slotHost[RT_SLOT_BVH_NODES]; const auto& T = c->slotHost[RT_SLOT_TRIANGLES];
        const bool dbgT = std::getenv("RT_DEBUG") != nullptr;
        auto tNow = [] { return std::chrono::steady_clock::now(); };
        auto tSec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        const auto tc0 = tNow();
        std::string e = compact_scene((const RefVertex*)V.data(), V.size() / 80, (const RefMat4*)M.data(), M.size() / 64,
                                      c->slotHost[RT_SLOT_MATERIALS].size() / 80, (const RefNode*)N.data(), N.size() / 48,
                                      (const RefTriangle*)T.data(), T.size() / 128, c->cs);
        if (!e.empty()) return fail(c, RT_ERR_SCENE, e);
        const auto tc1 = tNow();
        rt_status s;
        if ((s = upload(c, c->dNodes, c->cs.nodes.data(), c->cs.nodes.size() * sizeof(CNode)))) return s;
        if ((s = upload(c, c->dTris, c->cs.tris.data(), c->cs.tris.size() * sizeof(CTri)))) return s;
        if ((s = upload(c, c->dShade, c->cs.shade.data(), c->cs.shade.size() * sizeof(CShade)))) return s;
        if ((s = upload(c, c->dVertices, V.data(), V.size()))) return s;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        // the device has them now: the host keeps the node arrays (two-level derivation) and the counts
        c->cs.shade.clear(); c->cs.shadeUV.clear();
        c->cs.tris.clear();
        c->uvReady = false;
        if (dbgT) std::fprintf(stderr, "[rt] scene sync: compact layout derived in %.3f s, uploaded in %.3f s\n", tSec(tc0, tc1), tSec(tc1, tNow()));
        c->sceneDirty = false;
        c->tuneState = 0; c->tuneChoice = 0; c->shapeState = 0; c->shapeChoice = 0; c->tuneRateBase = 0.0;
        c->sceneKey = scene_key(c);
        if (c->tuneCache != 0) {
            TuneCache& T = tune_cache();
            std::lock_guard<std::mutex> g(T.mu);
            auto it = T.choice.find(c->sceneKey);
            if (it != T.choice.end()) { c->tuneState = 4; c->tuneChoice = it->second & 3; c->shapeState = 2; c->shapeChoice = (it->second >> 2) & 1; }
        }
        c->lateLogOff = false;      // (a new scene: its paths may be short again)
        c->wideReady = false; c->cs.wnodes.clear(); c->cs.wideDepth = 0;
    }
    if (want_wide(c) && !c->wideReady) {
        if (derive_wide(c->cs)) {
            rt_status s = upload(c, c->dWNodes, c->cs.wnodes.data(), c->cs.wnodes.size() * sizeof(WNode));
            if (s) return s;
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        c->wideReady = true;
    }
    if (c->materialsDirty) {
        const auto& M = c->slotHost[RT_SLOT_MATERIALS];
        // texture handles must name textures of this context
        for (size_t i = 0; i < M.size() / 80; i++) {
            const RefMaterial* m = (const RefMaterial*)M.data() + i;
            if ((m->hasBaseTexture == 1 && (m->baseColorTexture == 0 || m->baseColorTexture > c->textures.size())) ||
                (m->hasNormalTexture == 1 && (m->normalTexture == 0 || m->normalTexture > c->textures.size())) ||
                (m->hasMetallicRoughnessTexture == 1 && (m->metallicRoughnessTexture == 0 || m->metallicRoughnessTexture > c->textures.size())))
                return fail(c, RT_ERR_SCENE, "material " + std::to_string(i) + " names a texture handle rt_create_texture did not return");
        }
        rt_status s = upload(c, c->dMaterials, M.data(), M.size());
        if (s) return s;
        c->nMaterials = M.size() / 80;
        std::vector<CMat> cm(c->nMaterials);
        c->anyTextured = c->anyTransmissive = false;
        for (size_t i = 0; i < cm.size(); i++) {
            const RefMaterial* m = (const RefMaterial*)M.data() + i;
            for (int k = 0; k < 3; k++) cm[i].baseColor[k] = m->baseColorFactor[k];
            cm[i].transmission = m->transmissionFactor; cm[i].metallic = m->metallicFactor; cm[i].roughness = m->roughnessFactor;
            cm[i].ior = m->ior;
            cm[i].flags = (m->hasBaseTexture == 1 ? 1u : 0u) | (m->hasNormalTexture == 1 ? 2u : 0u) | (m->hasMetallicRoughnessTexture == 1 ? 4u : 0u);
            if (cm[i].flags) c->anyTextured = true;
            if (m->transmissionFactor > 0) c->anyTransmissive = true;      // the shader's own test (shader.comp:391); NaN is not > 0
        }
        if ((s = upload(c, c->dCMat, cm.data(), cm.size() * sizeof(CMat)))) return s;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->materialsDirty = false;
    }
    if (c->anyTextured && !c->uvReady) {
        // texture coordinates and vertex ids per triangle: only textured / normal-mapped materials read them
        const auto& V = c->slotHost[RT_SLOT_VERTICES]; const auto& T = c->slotHost[RT_SLOT_TRIANGLES];
        fill_shade_uv((const RefVertex*)V.data(), (const RefTriangle*)T.data(), T.size() / 128, c->cs);
        rt_status s = upload(c, c->dShadeUV, c->cs.shadeUV.data(), c->cs.shadeUV.size() * sizeof(CShadeUV));
        if (s) return s;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->cs.shadeUV.clear();
        c->uvReady = true;
    }
    if (c->lightsDirty) {
        const auto& L = c->slotHost[RT_SLOT_LIGHTS];
        rt_status s = upload(c, c->dLights, L.data(), L.size());
        if (s) return s;
        c->nLights = L.size() / 48;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->lightsDirty = false;
    }
    if (c->texturesDirty) {
        std::vector<TexDesc> d(c->textures.size());
        for (size_t i = 0; i < d.size(); i++) { d[i].rgba8 = (const uint8_t*)c->textures[i].data.p; d[i].width = c->textures[i].w; d[i].height = c->textures[i].h; }
        rt_status s = upload(c, c->dTexDescs, d.data(), d.size() * sizeof(TexDesc));
        if (s) return s;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->texturesDirty = false;
    }
    return RT_OK;
}

SceneView make_view(rt_ctx* c, bool exact) {
    SceneView S;
    std::memset(&S, 0, sizeof S);
    S.nodes = (const CNode*)c->dNodes.p; S.tris = (const CTri*)c->dTris.p; S.shade = (const CShade*)c->dShade.p;
    S.shadeUV = (const CShadeUV*)c->dShadeUV.p; S.cmat = (const CMat*)c->dCMat.p;
    S.materials = (const RefMaterial*)c->dMaterials.p; S.vertices = (const RefVertex*)c->dVertices.p;
    S.lights = (const RefLight*)c->dLights.p; S.textures = (const TexDesc*)c->dTexDescs.p;
    S.n_textures = (uint32_t)c->textures.size();
    S.rootChild = c->cs.rootChild;
    for (int k = 0; k < 3; k++) { S.rootLo[k] = c->cs.rootLo[k]; S.rootHi[k] = c->cs.rootHi[k]; }
    S.cullAbs = 1e-5f * c->cs.sceneScale;
    S.exactMode = exact ? 1 : 0;
    S.wnodes = (!exact && want_wide(c) && c->wideReady && !c->cs.wnodes.empty()) ? (const WNode*)c->dWNodes.p : nullptr;
    return S;
}

// The culled traversal is the reference's result only while the reference's own 64-entry stack cannot fill up (a push
// it drops changes what it renders, Q10) and while boxes contain what they lead to (Q3): otherwise the reference-order
// walk, which emulates that stack, is used whatever traversalMode says.
bool use_exact(rt_ctx* c) { return c->traversalMode == 1 || !c->cs.boxesContainTris || c->cs.maxDepth >= 63; }

// Launch geometry of the persistent traversal / shading kernels and the stack overflow area that goes with it.
uint32_t trace_blocks(rt_ctx* c, uint32_t nslots) {
    uint32_t slotBlocks = (nslots + RT_BLOCK - 1) / RT_BLOCK;
    uint32_t persistent = (uint32_t)c->numCUs * (uint32_t)c->blocksPerCU;
    return slotBlocks < persistent ? slotBlocks : persistent;
}
// Persistent grid of the refill kernels.  Six workgroups per CU are resident at once (80 registers: 6 waves per SIMD) and
// a kernel running by itself is fastest with exactly those; when passes overlap on several streams, four per CU leave
// room for the other streams' kernels and the frame is faster (C3 -5 %, C5 -7 %, C4 -4 % against 16; measured 2..16).
uint32_t refill_blocks(rt_ctx* c, uint32_t nItems, int nSets) {
    uint32_t itemBlocks = (nItems + RT_BLOCK - 1) / RT_BLOCK;
    const int perCU = c->refillBlocksPerCU > 0 ? (int)c->refillBlocksPerCU : (nSets > 1 ? 4 : 6);
    uint32_t most = (uint32_t)c->numCUs * (uint32_t)perCU;
    return std::max(1u, itemBlocks < most ? itemBlocks : most);
}
// Threads that may hold a traversal stack at one time = the largest grid any traversal kernel is launched with
// (batch kernels: trace_blocks; refill kernels: refill_blocks with 4, 6 or refillBlocksPerCU workgroups per CU).  The
// overflow area and its stride are sized from this, so that entry k of thread g never aliases another thread's.
uint32_t stack_blocks(rt_ctx* c, uint32_t nslots) {
    uint32_t slotBlocks = (nslots + RT_BLOCK - 1) / RT_BLOCK;
    const uint32_t perCU = (uint32_t)std::max(std::max((int)c->blocksPerCU, 6), (int)c->refillBlocksPerCU);
    uint32_t most = (uint32_t)c->numCUs * perCU;
    return std::max(1u, slotBlocks < most ? slotBlocks : most);
}
uint32_t overflow_depth(rt_ctx* c) {
    // ordered traversal keeps at most one pending sibling per level: depth <= tree depth (two-level nodes: three per level)
    int need = (int)c->cs.maxDepth + 1 - RT_LDS_STACK;
    if (!c->cs.wnodes.empty()) need = std::max(need, 3 * (int)c->cs.wideDepth + 1 - RT_LDS_STACK);
    if (need < 0) need = 0;
    if (need > RT_SPILL_STACK) need = RT_SPILL_STACK;
    return (uint32_t)need;
}

void release_paths(rt_ctx* c) {
    for (DevBuf& b : c->arena) b.release();
    for (auto& ps : c->set) { ps.counters.release(); ps.overflow.release(); }
}

// The contribution log of a pass: planes [0, dense) hold an entry per slot; the bounces from `dense` on, which few paths
// reach when the bounce limit is high, hold `lateCap` entries each, addressed by queue position (PathBuffers, k_shade).
struct LogShape { uint32_t dense, lateBounces, lateCap; };
LogShape log_shape(rt_ctx* c, uint32_t nslots, uint32_t maxBounces) {
    LogShape g;
    const uint32_t maxB = std::max<uint32_t>(1, maxBounces);
    int dense = c->logDenseBounces;
    if (dense <= 0) dense = maxB > 8 ? 4 : (int)maxB;      // automatic: two parts only where the log dominates the path state
    if (c->lateLogOff) dense = (int)maxB;
    g.dense = std::min<uint32_t>((uint32_t)dense, maxB);
    g.lateBounces = maxB - g.dense;
    const uint64_t cap = ((uint64_t)nslots * (uint64_t)std::min(std::max(c->lateLogPercent, 1), 100) / 100 + 63) / 64 * 64;
    g.lateCap = g.lateBounces ? (uint32_t)std::min<uint64_t>(std::max<uint64_t>(cap, 64), (uint64_t)nslots + 63) : 0u;
    return g;
}

// Bytes of one set's part of every arena array (multiples of 256, so that every set's part is aligned).
struct SetBytes { size_t b[rt_ctx::A_COUNT]; };
SetBytes set_bytes(rt_ctx* c, uint32_t nslots, uint32_t lightSlots, uint32_t maxBounces) {
    SetBytes sb;
    const size_t n = nslots;
    const LogShape g = log_shape(c, nslots, maxBounces);
    sb.b[rt_ctx::A_RAY0] = sb.b[rt_ctx::A_RAY1] = n * 32;
    sb.b[rt_ctx::A_RAYS0] = sb.b[rt_ctx::A_RAYS1] = n * 16;
    sb.b[rt_ctx::A_HIT] = n * 8;
    sb.b[rt_ctx::A_SHADOWREC] = n * 16;
    sb.b[rt_ctx::A_SHADELOC] = n * 16;
    sb.b[rt_ctx::A_CONTRIB] = n * sizeof(LogEntry) * lightSlots * g.dense;
    sb.b[rt_ctx::A_DEPTH] = n * 4;
    sb.b[rt_ctx::A_LATE] = g.lateBounces ? (size_t)g.lateBounces * lightSlots * g.lateCap * sizeof(LogEntry) : 0;
    sb.b[rt_ctx::A_LATENEXT] = g.lateBounces ? (size_t)g.lateBounces * g.lateCap * 4 : 0;
    sb.b[rt_ctx::A_LATEHEAD] = g.lateBounces ? n * 4 : 0;
    for (size_t& x : sb.b) x = (x + 255) / 256 * 256;
    return sb;
}

// Path buffers for `nSets` sets of `nslots` slots; `alsoSets` x `alsoSlots` (0: none) is a second pass shape the arena has
// to hold as well (the candidate of the pass-shape trial), so that switching between the two allocates nothing.
rt_status ensure_paths(rt_ctx* c, int nSets, uint32_t nslots, uint32_t slotsPerSample, uint32_t lightSlots, uint32_t maxBounces,
                       int alsoSets = 0, uint32_t alsoSlots = 0) {
    rt_status s;
    const SetBytes sb = set_bytes(c, nslots, lightSlots, maxBounces);
    SetBytes also{};
    if (alsoSets > 0) also = set_bytes(c, alsoSlots, lightSlots, maxBounces);
    for (int a = 0; a < rt_ctx::A_COUNT; a++) {
        size_t need = sb.b[a] * (size_t)nSets;
        if (alsoSets > 0) need = std::max(need, also.b[a] * (size_t)alsoSets);
        if (need && (s = ensure(c, c->arena[a], need))) return s;
    }
    const uint32_t mostSlots = std::max(nslots, alsoSets > 0 ? alsoSlots : 0u);
    for (int k = 0; k < std::max(nSets, alsoSets); k++) {
        if ((s = ensure(c, c->set[k].overflow, (size_t)overflow_depth(c) * stack_blocks(c, mostSlots) * RT_BLOCK * sizeof(uint2)))) return s;
        if ((s = ensure(c, c->set[k].counters, (size_t)(2 * (maxBounces + 1)) * 4))) return s;
    }
    if ((s = ensure(c, c->accum, (size_t)slotsPerSample * 16))) return s;
    if ((s = ensure(c, c->dStats, sizeof(DeviceStats)))) return s;
    return RT_OK;
}

PathBuffers make_paths(rt_ctx* c, int which, uint32_t nslots, uint32_t slotsPerSample, uint32_t lightSlots, uint32_t maxBounces) {
    PathBuffers P;
    const SetBytes sb = set_bytes(c, nslots, lightSlots, maxBounces);
    auto part = [&](int a) -> void* { return c->arena[a].p ? (char*)c->arena[a].p + sb.b[a] * (size_t)which : nullptr; };
    rt_ctx::PathSet& ps = c->set[which];
    P.ray[0] = (float4*)part(rt_ctx::A_RAY0); P.ray[1] = (float4*)part(rt_ctx::A_RAY1);
    P.rayS[0] = (float4*)part(rt_ctx::A_RAYS0); P.rayS[1] = (float4*)part(rt_ctx::A_RAYS1);
    P.hit = (float2*)part(rt_ctx::A_HIT); P.shadowRec = (float4*)part(rt_ctx::A_SHADOWREC); P.shadeLoc = (float4*)part(rt_ctx::A_SHADELOC);
    P.contrib = (LogEntry*)part(rt_ctx::A_CONTRIB); P.depth = (uint32_t*)part(rt_ctx::A_DEPTH); P.accum = (float4*)c->accum.p;
    P.counters = (uint32_t*)ps.counters.p;
    P.nslots = nslots; P.slotsPerSample = slotsPerSample; P.lightSlots = lightSlots; P.grpShift = 0;
    const LogShape g = log_shape(c, nslots, maxBounces);
    P.late = g.lateBounces ? (LogEntry*)part(rt_ctx::A_LATE) : nullptr; P.lateNext = g.lateBounces ? (uint32_t*)part(rt_ctx::A_LATENEXT) : nullptr;
    P.lateHead = g.lateBounces ? (uint32_t*)part(rt_ctx::A_LATEHEAD) : nullptr;
    P.denseBounces = g.lateBounces ? g.dense : 0xffffffffu; P.lateCap = g.lateCap;
    P.overflow = (uint2*)ps.overflow.p; P.overflowStride = stack_blocks(c, nslots) * RT_BLOCK; P.overflowDepth = overflow_depth(c);
    return P;
}

template <bool PRIMARY, typename... Args>
void launch_extend(bool exact, bool count, dim3 g, hipStream_t st, Args... a) {
    if (exact) { if (count) k_extend<true, true, PRIMARY><<<g, RT_BLOCK, 0, st>>>(a...); else k_extend<true, false, PRIMARY><<<g, RT_BLOCK, 0, st>>>(a...); }
    else { if (count) k_extend<false, true, PRIMARY><<<g, RT_BLOCK, 0, st>>>(a...); else k_extend<false, false, PRIMARY><<<g, RT_BLOCK, 0, st>>>(a...); }
}
template <typename... Args>
void launch_shadow(bool exact, bool count, dim3 g, hipStream_t st, Args... a) {
    if (exact) { if (count) k_shadow<true, true><<<g, RT_BLOCK, 0, st>>>(a...); else k_shadow<true, false><<<g, RT_BLOCK, 0, st>>>(a...); }
    else { if (count) k_shadow<false, true><<<g, RT_BLOCK, 0, st>>>(a...); else k_shadow<false, false><<<g, RT_BLOCK, 0, st>>>(a...); }
}

rt_status render_region(rt_ctx* c, uint32_t x, uint32_t y, uint32_t w, uint32_t h, uint32_t r0, uint32_t r1) {
    if (!c) return RT_ERR_INVALID;
    if (c->width == 0 || c->height == 0) return fail(c, RT_ERR_NOT_READY, "rt_resize has not been called");
    if (w == 0 || h == 0 || x >= c->width || w > c->width - x || y >= c->height || h > c->height - y)      // (no 32-bit wrap in x + w)
        return fail(c, RT_ERR_INVALID, "region outside the image");
    if (c->samplesPerPixel < 1 || c->maxBounces < 0) return fail(c, RT_ERR_INVALID, "samplesPerPixel must be >= 1 and maxBounces >= 0");
    if (r0 > r1 || r1 > (uint32_t)c->samplesPerPixel) return fail(c, RT_ERR_INVALID, "sample range outside [0, samplesPerPixel]");
    rt_status s = sync_scene(c);
    if (s) return s;
    if ((uint64_t)c->numLights > c->nLights) return fail(c, RT_ERR_INVALID, "numLights exceeds the light buffer (slot 5)");
    auto t0 = std::chrono::steady_clock::now();
    const uint64_t allocationsBefore = c->allocations;

    FrameParams F;
    std::memset(&F, 0, sizeof F);
    F.width = (int)c->width; F.height = (int)c->height;
    F.rx = (int)x; F.ry = (int)y; F.rw = (int)w; F.rh = (int)h;
    F.tilesX = (int)((w + 7) / 8);
    const uint32_t tilesY = (h + 7) / 8;
    const uint32_t sps = (uint32_t)F.tilesX * tilesY * 64u;            // slots per sample
    F.numLights = c->numLights < 0 ? 0 : c->numLights; F.hasSun = c->hasSun;
    for (int l = 0; l < F.numLights && (size_t)(l + 1) * 48 <= c->slotHost[RT_SLOT_LIGHTS].size(); l++)
        if (((const RefLight*)c->slotHost[RT_SLOT_LIGHTS].data())[l].lightType == 0u) F.anyPoint = 1;
    std::memcpy(F.lights, c->slotHost[RT_SLOT_LIGHTS].data(), std::min<size_t>(sizeof F.lights, c->slotHost[RT_SLOT_LIGHTS].size()));
    for (int k = 0; k < 3; k++) { F.camPos[k] = c->camPos[k]; F.camForward[k] = c->camForward[k]; F.camRight[k] = c->camRight[k]; F.camUp[k] = c->camUp[k]; }
    F.aspect = c->aspect;
    F.tanHalfFov = tanf((c->fov * 0.017453292519943295f) / 2.0f);   // tan(radians(fov) / 2.0), shader.comp:321
    F.samplesPerPixel = c->samplesPerPixel; F.maxBounces = c->maxBounces;
    F.gridN = (int)ceilf(sqrtf((float)c->samplesPerPixel));            // shader.comp:309-310
    F.initialStrength = 1.f / (float)c->samplesPerPixel;               // shader.comp:307
    F.strengthCutoff = .01f / (float)c->samplesPerPixel;               // shader.comp:306

    // Samples in flight per pass: enough rays per launch to amortise the latency-bound tail of the traversal
    // kernels, bounded by a slot budget.
    const uint32_t maxB = (uint32_t)c->maxBounces;
    // Samples of a pixel per wave (slot_split): 16 (of four pixels) — or all 64 of one pixel where the BVH is far larger than
    // the L2s: there the traversal kernels gain more from rays that start at one point than k_resolve loses.
    const int sampleGroup = c->sampleGroup < 0 ? (bvh_bytes(c) > kBigBvhBytes ? 6 : 4) : std::min(c->sampleGroup, 6);
    // The samples per pass of a frame rendered on `wantStreams` streams with at most `sppCap` samples per pass.
    auto plan_pass = [&](int wantStreams, int sppCap) -> uint32_t {
        uint32_t S_ = (uint32_t)(sppCap < 1 ? 1 : sppCap);
        // Slot budget: all buffer sets together may take `pathMemPercent` of the device's memory (the scene, the
        // images and other tenants keep the rest), and a set at most 2^30 slots.
        {
            const LogShape g = log_shape(c, 1u << 20, maxB);      // (shape only: the dense / late split does not depend on the slot count)
            const uint64_t L = (uint64_t)std::max(1, F.numLights);
            const uint64_t lateBytes = g.lateBounces ? 4 + ((uint64_t)g.lateBounces * (sizeof(LogEntry) * L + 4) * (uint64_t)std::min(std::max(c->lateLogPercent, 1), 100) + 99) / 100 : 0;
            const uint64_t bytesPerSlot = 2 * 32 + 2 * 16 + 8 + 16 + 16 + 4 + (uint64_t)sizeof(LogEntry) * L * g.dense + lateBytes;
            const uint64_t wantSets = (uint64_t)std::min<int>(std::max(1, wantStreams), rt_ctx::kMaxSets);
            uint64_t slotBudget = std::min<uint64_t>(1ull << 30, (c->deviceBytes / 100 * (uint64_t)std::min(std::max(c->pathMemPercent, 1), 90)) / (wantSets * bytesPerSlot));
            if ((uint64_t)S_ * sps > slotBudget) S_ = (uint32_t)std::max<uint64_t>(1, slotBudget / sps);
        }
        if (S_ > r1 - r0) S_ = std::max<uint32_t>(1, r1 - r0);
        // A short sample range (one rank's share of a frame split over several GPUs) is cut into at least `wantStreams`
        // passes of >= 16 samples, so that the passes still overlap on the streams.
        if (c->autoPass != 0 && wantStreams > 1) {
            const uint32_t want = (r1 - r0 + (uint32_t)wantStreams - 1) / (uint32_t)wantStreams;
            if (want < S_) S_ = std::max<uint32_t>(std::min<uint32_t>(16u, S_), want);
        }
        // Whole sample groups (slot_split: 2^sampleGroup samples of a pixel share a wave), and as many passes as fill the
        // streams evenly: 6 passes on 4 streams leave two streams idle for the last third of the frame.
        {
            const uint32_t n = r1 - r0;
            uint32_t q = 1u << (uint32_t)sampleGroup;
            while (q > 1u && q > S_) q >>= 1;
            S_ -= S_ % q;
            const uint32_t streams = (uint32_t)std::min<int>(std::max(1, wantStreams), rt_ctx::kMaxSets);
            uint32_t nPasses = (n + S_ - 1) / S_;
            if (c->autoPass != 0 && nPasses > streams && nPasses % streams != 0u) {
                nPasses += streams - nPasses % streams;
                uint32_t even = (n + nPasses - 1) / nPasses;
                even = (even + q - 1) / q * q;
                if (even >= 1u && even <= S_) S_ = even;
            }
        }
        return S_;
    };
    const bool exact = use_exact(c);
    const bool count = c->countVisits != 0;
    // a frame large enough to be a measurement (config 1's 512 x 512 x 64 is)
    const bool sizeable = (double)w * (double)h * (double)(r1 - r0) >= 1.0e7;
    const bool tunable = !exact && !count && c->refill < 0 && c->kernelChoice < 0 && sizeable;
    // Pass shape (see rt_ctx::shapeChoice): the caller's, a pinned one (bit 2 of kernelChoice), the measured one, or — on
    // the frames after the kernel trial — the candidate being measured.
    const int halfStreams = std::max(1, (int)c->passStreams / 2), wideSpp = 2 * std::max(1, (int)c->samplesPerPass);
    bool shapeTrial = false, shapeDecidedNow = false;
    int shape = 0;
    if (!shape_user(c) && c->passStreams >= 2) {
        if (c->kernelChoice >= 0) shape = (c->kernelChoice >> 2) & 1;
        else if (c->shapeState == 2) shape = c->shapeChoice;
        else if (tunable && c->tuneState >= 4 && c->tuneRateBase > 0.0) {
            // (a candidate that cuts the frame exactly as the default does needs no measurement)
            const uint32_t s0 = plan_pass(c->passStreams, c->samplesPerPass), s1 = plan_pass(halfStreams, wideSpp);
            const uint32_t n0 = std::min<uint32_t>((r1 - r0 + s0 - 1) / s0, (uint32_t)c->passStreams), n1 = std::min<uint32_t>((r1 - r0 + s1 - 1) / s1, (uint32_t)halfStreams);
            if (s0 == s1 && n0 == n1) { c->shapeState = 2; c->shapeChoice = 0; shapeDecidedNow = true; }
            else { shapeTrial = true; shape = 1; c->shapeState = 1; }
        }
    }
    const int effStreams = shape ? halfStreams : (int)c->passStreams;
    uint32_t S_ = plan_pass(effStreams, shape ? wideSpp : (int)c->samplesPerPass);
    // While the shape is undecided the arena is sized for both, so that trying the other one allocates nothing.
    const bool bothShapes = !shape_user(c) && c->passStreams >= 2 && c->kernelChoice < 0 && c->shapeState < 2 && tunable;
    const int otherStreams = shape ? (int)c->passStreams : halfStreams;
    const uint32_t otherS = bothShapes ? plan_pass(otherStreams, shape ? (int)c->samplesPerPass : wideSpp) : 0u;
    const uint32_t lightSlots = (uint32_t)std::max(1, F.numLights);

    // Path buffers: `passStreams` sets of S_ samples.  When the device cannot hold them (another tenant, a smaller
    // part), fall back first to fewer streams, then to fewer samples in flight — the image does not depend on either.
    uint32_t nslots = 0;
    int nSets = 1;
    PathBuffers Pset[rt_ctx::kMaxSets];
    int capSets = std::min<int>(std::max(1, effStreams), rt_ctx::kMaxSets);
    bool roomForBoth = bothShapes;
    for (;;) {
        nslots = S_ * sps;
        const uint32_t nPasses = (r1 - r0 + S_ - 1) / std::max(1u, S_);
        nSets = (int)std::max(1u, std::min<uint32_t>((uint32_t)capSets, nPasses));
        int alsoSets = 0; uint32_t alsoSlots = 0;
        if (roomForBoth && otherS > 0) {
            alsoSlots = otherS * sps;
            alsoSets = (int)std::max(1u, std::min<uint32_t>((uint32_t)std::min<int>(std::max(1, otherStreams), rt_ctx::kMaxSets), (r1 - r0 + otherS - 1) / otherS));
        }
        if (!(s = ensure_paths(c, nSets, nslots, sps, lightSlots, maxB, alsoSets, alsoSlots))) break;
        (void)hipGetLastError();
        release_paths(c);
        if (roomForBoth) { roomForBoth = false; continue; }      // first without the room for the other shape
        if (nSets > 1) { capSets = nSets - 1; continue; }         // then fewer streams
        if (S_ <= 1) return s;
        S_ = std::max(1u, S_ / 2);                                // then fewer samples in flight
    }
    if (bothShapes && !roomForBoth) { c->shapeState = 2; c->shapeChoice = 0; shapeTrial = false; }      // (no room to hold both shapes: the default stays)
    for (int k = 0; k < nSets; k++) Pset[k] = make_paths(c, k, nslots, sps, lightSlots, maxB);
    PathBuffers P = Pset[0];
    SceneView S = make_view(c, exact);
    DeviceStats* dst = (DeviceStats*)c->dStats.p;
    hipStream_t st = c->stream;

    HIPCHK(c, hipMemsetAsync(dst, 0, sizeof(DeviceStats), st));
    // colour image: zero outside the region
    HIPCHK(c, hipMemsetAsync(c->img[0].p, 0, (size_t)c->width * c->height * 16, st));
    const bool aov = (r0 == 0);
    float4* iAlb = aov ? (float4*)c->img[1].p : nullptr;
    float4* iNor = aov ? (float4*)c->img[2].p : nullptr;
    float4* iDep = aov ? (float4*)c->img[3].p : nullptr;
    int32_t* pTri = aov ? (int32_t*)c->primTri.p : nullptr;
    float* pHit = aov ? (float*)c->primHit.p : nullptr;
    // AOV images: zero everywhere, then written by the call whose sample range contains r = 0 (so that partial
    // renders of disjoint regions / sample ranges can be merged with one sum)
    for (int i = 1; i < 4; i++) HIPCHK(c, hipMemsetAsync(c->img[i].p, 0, (size_t)c->width * c->height * 16, st));
    if (aov) {
        HIPCHK(c, hipMemsetAsync(c->primTri.p, 0xFF, (size_t)c->width * c->height * 4, st));
        HIPCHK(c, hipMemsetAsync(c->primHit.p, 0, (size_t)c->width * c->height * 12, st));
    }
    const uint32_t pixBlocks = (sps + RT_BLOCK - 1) / RT_BLOCK;
    k_init_frame<<<pixBlocks, RT_BLOCK, 0, st>>>(F, P, iAlb, iNor, iDep);

    const uint32_t traceBlocks = trace_blocks(c, nslots);
    const bool bigScene = bvh_bytes(c) > kBigBvhBytes;
    // Which traversal kernels: bit 0 = refill kernel for closest-hit rays, bit 1 = for shadow rays.
    static const int kTuneTrial[4] = {1, 3, 0, 2};
    // until measured: the refill kernel for closest-hit rays (the faster one on every scene measured so far), and for
    // shadow rays too where traversal is bound by miss latency
    const int comboDefault = bigScene ? 3 : 1;
    const bool trial = tunable && c->tuneState < 4;
    const int combo = exact ? 0 : c->refill > 0 ? 3 : c->refill == 0 ? 0 : c->kernelChoice >= 0 ? (c->kernelChoice & 3) : c->tuneState >= 4 ? c->tuneChoice
                    : trial ? kTuneTrial[c->tuneState] : comboDefault;
    const bool refillExtendAll = (combo & 1) != 0, refillShadow = (combo & 2) != 0;
    const bool refillExtend = refillExtendAll;
    const bool wide = S.wnodes != nullptr;
    const int refillIdle = std::min(64, std::max(1, (int)c->refillIdle));
    const int refillNodeMin = std::min(64, std::max(1, (int)c->refillNodeMin)) | (std::min(64, std::max(0, (int)c->refillTriMin)) << 8) |
                              (std::min(64, std::max(0, (int)c->refillNodeLow)) << 16) | ((c->refillDefer != 0 ? 1 : 0) << 24);
    hipStream_t streams[rt_ctx::kMaxSets];
    streams[0] = c->stream;
    for (int k = 1; k < rt_ctx::kMaxSets; k++) streams[k] = c->extra[k];
    if (nSets > 1) {
        HIPCHK(c, hipEventRecord(c->evInit, st));                 // images + accumulators initialised
        for (int k = 1; k < nSets; k++) HIPCHK(c, hipStreamWaitEvent(streams[k], c->evInit, 0));
    }
    // Optional per-kernel-class timing with HIP events on the launch stream (ring of pairs, drained lazily).
    const bool timeKernels = c->timeKernels != 0;
    double msAcc[4] = {0.0, 0.0, 0.0, 0.0};     // 0 closest hit of bounce rays, 1 shading, 2 shadow rays, 3 closest hit of camera rays
    uint64_t nLaunch[4] = {0, 0, 0, 0};
    size_t evIdx = 0;
    if (timeKernels && c->evRing.empty()) {
        c->evRing.resize(1024);
        for (auto& e : c->evRing) { HIPCHK(c, hipEventCreate(&e.a)); HIPCHK(c, hipEventCreate(&e.b)); }
    }
    auto evDrain = [&](rt_ctx::EvPair& e) {
        if (!e.used) return;
        float ms = 0.f;
        (void)hipEventSynchronize(e.b);
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) msAcc[e.kind] += ms;
        e.used = false;
    };
    auto evBegin = [&](int kind) -> rt_ctx::EvPair* {
        nLaunch[kind]++;
        if (!timeKernels) return nullptr;
        rt_ctx::EvPair& e = c->evRing[evIdx++ % c->evRing.size()];
        evDrain(e);
        e.kind = kind; e.used = true;
        (void)hipEventRecord(e.a, st);
        return &e;
    };
    auto evEnd = [&](rt_ctx::EvPair* e) { if (e) (void)hipEventRecord(e->b, st); };
    uint32_t passNo = 0;
    uint64_t nPairs = 0;

    for (uint32_t rBase = r0; rBase < r1; rBase += S_) {
        const uint32_t inPass = std::min(S_, r1 - rBase);
        // consecutive passes rotate over the streams and buffer sets; k_resolve alone is ordered
        const int ks = (int)(passNo % (uint32_t)nSets);
        P = Pset[ks];
        st = streams[ks];
        uint32_t* qcount = P.counters;
        uint32_t* scount = P.counters + (maxB + 1);
        PathBuffers Pp = P;
        Pp.nslots = inPass * sps;                  // a short last pass uses its own (smaller) log strides
        {   // samples per group of 64 slots (slot_split): the largest power of two <= 2^sampleGroup that divides the pass
            uint32_t sh = (uint32_t)sampleGroup;
            while (sh > 0 && (inPass & ((1u << sh) - 1u)) != 0u) sh--;
            Pp.grpShift = sh;
        }
        const uint32_t passBlocks = (Pp.nslots + RT_BLOCK - 1) / RT_BLOCK;
        const uint32_t tb = std::min(traceBlocks, passBlocks);
        HIPCHK(c, hipMemsetAsync(P.counters, 0, (size_t)(2 * (maxB + 1)) * 4, st));
        HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)qcount, (int)Pp.nslots, 1, st));   // bounce 0: every slot carries a path
        const bool sun1 = F.numLights == 1 && F.lights[0].lightType == 1u && c->sunKernel;
        // Shadow rays of bounce b and closest-hit rays of bounce b + 1 both come out of k_shade(b) and do not depend on
        // each other: they are traced by one launch (k_trace_pair), which costs the slower of the two instead of their sum.
        const bool pairs = !exact && !count && c->fusePairs != 0 && F.numLights > 0;
        bool extendDone = false;
        for (uint32_t b = 0; b < maxB; b++) {
            const float4* qin = b == 0 ? nullptr : P.ray[b & 1];       // bounce 0: entry i is slot i, camera rays are regenerated
            rt_ctx::EvPair* e = nullptr;
            if (!extendDone) {
                e = evBegin(b == 0 ? 3 : 0);
                if (refillExtend && b > 0 && (int)b >= c->refillFromBounce) {      // camera rays are coherent: the batch kernel is the better one there
                    const dim3 rg(refill_blocks(c, Pp.nslots, nSets));
                    if (wide) {
                        if (count) k_extend_refill<true, false, true><<<rg, RT_BLOCK, 0, st>>>(S, F, Pp, qin, (const uint32_t*)(qcount + b), (int)rBase, 0u, refillIdle, refillNodeMin, dst);
                        else k_extend_refill<false, false, true><<<rg, RT_BLOCK, 0, st>>>(S, F, Pp, qin, (const uint32_t*)(qcount + b), (int)rBase, 0u, refillIdle, refillNodeMin, dst);
                    }
                    else if (count) k_extend_refill<true, false><<<rg, RT_BLOCK, 0, st>>>(S, F, Pp, qin, (const uint32_t*)(qcount + b), (int)rBase, 0u, refillIdle, refillNodeMin, dst);
                    else k_extend_refill<false, false><<<rg, RT_BLOCK, 0, st>>>(S, F, Pp, qin, (const uint32_t*)(qcount + b), (int)rBase, 0u, refillIdle, refillNodeMin, dst);
                }
                else if (b == 0) launch_extend<true>(exact, count, dim3(tb), st, S, F, Pp, (const float4*)nullptr, (const uint32_t*)qcount, (int)rBase, inPass * w * h, dst);
                else launch_extend<false>(exact, count, dim3(tb), st, S, F, Pp, qin, (const uint32_t*)(qcount + b), (int)rBase, 0u, dst);
                evEnd(e);
            }
            extendDone = false;
            e = evBegin(1);
            {
                const bool many = F.numLights > RT_MAX_LIGHTS;
                const bool lateLog = Pp.denseBounces < maxB;      // this frame's log has a compact part (log_shape)
                auto shadeK = lateLog
                            ? (c->anyTextured ? (many ? k_shade<true, true, true, false, true> : k_shade<true, true, false, false, true>)
                               : c->anyTransmissive ? (many ? k_shade<false, true, true, false, true> : sun1 ? k_shade<false, true, false, true, true> : k_shade<false, true, false, false, true>)
                                                    : (many ? k_shade<false, false, true, false, true> : sun1 ? k_shade<false, false, false, true, true> : k_shade<false, false, false, false, true>))
                            : (c->anyTextured ? (many ? k_shade<true, true, true, false> : k_shade<true, true, false, false>)
                               : c->anyTransmissive ? (many ? k_shade<false, true, true, false> : sun1 ? k_shade<false, true, false, true> : k_shade<false, true, false, false>)
                                                    : (many ? k_shade<false, false, true, false> : sun1 ? k_shade<false, false, false, true> : k_shade<false, false, false, false>));
                shadeK<<<tb, RT_BLOCK, 0, st>>>(S, F, Pp, (const uint32_t*)(qcount + b), qcount + b + 1,
                                                 scount + b, (int)rBase, (int)b, iAlb, iNor, iDep, pTri, pHit, dst);
            }
            evEnd(e);
            if (F.numLights > 0) {
                if (pairs && b + 1 < maxB) {
                    const uint32_t rg = refill_blocks(c, Pp.nslots, nSets);
                    const bool refillExtend = refillExtendAll && (int)(b + 1) >= c->refillFromBounce;      // (this launch's closest-hit rays are bounce b + 1's)
                    const uint32_t gE = refillExtend ? rg : tb; uint32_t gS = refillShadow ? rg : tb;
                    const float4* qnext = P.ray[(b + 1) & 1];
                    const uint32_t* cE = qcount + b + 1;
                    const uint32_t* cS = scount + b;
                    const bool packet = sun1 && (int)b < packet_bounces(c);
                    const int shd = packet ? 4 : refillShadow ? (sun1 ? 3 : 2) : (sun1 ? 1 : 0);
                    if (packet) gS = tb;
                    const dim3 g(std::max(gE, gS));
                    e = evBegin(0);
#define RT_PAIR(EXT, SHD) k_trace_pair<EXT, SHD><<<g, RT_BLOCK, 0, st>>>(S, F, Pp, qnext, cE, (int)rBase, cS, (int)b, gE, gS, refillIdle, refillNodeMin, dst)
#define RT_PAIRW(EXT, SHD) k_trace_pair<EXT, SHD, true><<<g, RT_BLOCK, 0, st>>>(S, F, Pp, qnext, cE, (int)rBase, cS, (int)b, gE, gS, refillIdle, refillNodeMin, dst)
                    if (wide && (refillExtend || (refillShadow && shd != 4))) {
                        if (refillExtend) { if (shd == 0) RT_PAIRW(1, 0); else if (shd == 1) RT_PAIRW(1, 1); else if (shd == 2) RT_PAIRW(1, 2); else if (shd == 3) RT_PAIRW(1, 3); else RT_PAIRW(1, 4); }
                        else { if (shd == 2) RT_PAIRW(0, 2); else RT_PAIRW(0, 3); }
                    }
                    else if (refillExtend) { if (shd == 0) RT_PAIR(1, 0); else if (shd == 1) RT_PAIR(1, 1); else if (shd == 2) RT_PAIR(1, 2); else if (shd == 3) RT_PAIR(1, 3); else RT_PAIR(1, 4); }
                    else { if (shd == 0) RT_PAIR(0, 0); else if (shd == 1) RT_PAIR(0, 1); else if (shd == 2) RT_PAIR(0, 2); else if (shd == 3) RT_PAIR(0, 3); else RT_PAIR(0, 4); }
#undef RT_PAIR
#undef RT_PAIRW
                    evEnd(e);
                    nPairs++;
                    extendDone = true;
                    continue;
                }
                e = evBegin(2);
                if (!exact && !count && sun1 && (int)b < packet_bounces(c))
                    k_shadow_sun_packet<<<tb, RT_BLOCK, 0, st>>>(S, F, Pp, (const uint32_t*)(scount + b), (int)b);
                else if (refillShadow) {
                    const dim3 rg(refill_blocks(c, Pp.nslots, nSets));
                    if (wide) {
                        if (count) k_shadow_refill<true, false, true><<<rg, RT_BLOCK, 0, st>>>(S, F, Pp, (const uint32_t*)(scount + b), (int)b, refillIdle, refillNodeMin, dst);
                        else if (sun1) k_shadow_refill<false, true, true><<<rg, RT_BLOCK, 0, st>>>(S, F, Pp, (const uint32_t*)(scount + b), (int)b, refillIdle, refillNodeMin, dst);
                        else k_shadow_refill<false, false, true><<<rg, RT_BLOCK, 0, st>>>(S, F, Pp, (const uint32_t*)(scount + b), (int)b, refillIdle, refillNodeMin, dst);
                    }
                    else if (count) k_shadow_refill<true, false><<<rg, RT_BLOCK, 0, st>>>(S, F, Pp, (const uint32_t*)(scount + b), (int)b, refillIdle, refillNodeMin, dst);
                    else if (sun1) k_shadow_refill<false, true><<<rg, RT_BLOCK, 0, st>>>(S, F, Pp, (const uint32_t*)(scount + b), (int)b, refillIdle, refillNodeMin, dst);
                    else k_shadow_refill<false, false><<<rg, RT_BLOCK, 0, st>>>(S, F, Pp, (const uint32_t*)(scount + b), (int)b, refillIdle, refillNodeMin, dst);
                }
                else if (!exact && !count && sun1)
                    k_shadow_sun<<<tb, RT_BLOCK, 0, st>>>(S, F, Pp, (const uint32_t*)(scount + b), (int)b);
                else launch_shadow(exact, count, dim3(tb), st, S, F, Pp, (const uint32_t*)(scount + b), (int)b, dst);
                evEnd(e);
            }
        }
        // a launch that was refused (bad configuration, out of resources) is reported with its pass, not at the frame's end
        {
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess) {
                for (int k = 0; k < nSets; k++) (void)hipStreamSynchronize(streams[k]);
                return fail(c, RT_ERR_HIP, "kernel launch failed in pass " + std::to_string(passNo) + " (samples from " + std::to_string(rBase) +
                                           ", " + std::to_string(maxB) + " bounces of k_extend / k_shade / k_shadow): " + hipGetErrorString(le));
            }
        }
        // the pixel sum runs over samples in order (shader.comp:312): this pass's replay follows the previous pass's
        if (nSets > 1 && passNo > 0) HIPCHK(c, hipStreamWaitEvent(st, c->evResolve[(ks + nSets - 1) % nSets], 0));
        // (maxBounces == 0: nothing was shaded, so the per-slot depths k_resolve reads were never written)
        if (maxB > 0) k_resolve<<<pixBlocks, RT_BLOCK, 0, st>>>(Pp, (int)inPass, (int)maxB, F.hasSun, F.initialStrength);
        if (nSets > 1) HIPCHK(c, hipEventRecord(c->evResolve[ks], st));
        passNo++;
    }
    k_finalize<<<pixBlocks, RT_BLOCK, 0, st>>>(F, P, (float4*)c->img[0].p, aov ? 1.f : 0.f);     // on the last pass's stream, after its replay
    HIPCHK(c, hipGetLastError());
    for (int k = 0; k < nSets; k++) {
        const hipError_t se = hipStreamSynchronize(streams[k]);
        if (se != hipSuccess)      // a fault inside a kernel surfaces here: say which stream's passes (k, k + nSets, ...) it was in
            return fail(c, RT_ERR_HIP, "a kernel of the passes on stream " + std::to_string(k) + " of " + std::to_string(nSets) + " failed: " + hipGetErrorString(se));
    }
    if (timeKernels) for (auto& e : c->evRing) evDrain(e);

    DeviceStats hs;
    HIPCHK(c, hipMemcpy(&hs, dst, sizeof hs, hipMemcpyDeviceToHost));
    if (hs.late_overflow != 0ull && !c->lateLogOff) {
        // more paths reached a late bounce than the compact part of the log holds: the frame is incomplete.  Render it
        // again with a dense log, and keep that for this context (a scene whose paths live long).
        c->lateLogOff = true;
        c->lateLogFallbacks++;
        return render_region(c, x, y, w, h, r0, r1);
    }
    auto t1 = std::chrono::steady_clock::now();
    std::memset(&c->stats, 0, sizeof c->stats);
    c->stats.rays_primary = hs.rays_primary; c->stats.rays_bounce = hs.rays_bounce; c->stats.rays_shadow = hs.rays_shadow;
    c->stats.rays_shadow_skipped = hs.rays_shadow_skipped; c->stats.node_visits = hs.node_visits; c->stats.tri_tests = hs.tri_tests;
    c->stats.ms_total = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (trial) {
        const double rays = (double)(hs.rays_primary + hs.rays_bounce + hs.rays_shadow);
        if (rays >= 3.0e7 && c->stats.ms_total > 0.0 && c->allocations == allocationsBefore) {
            c->tuneRate[c->tuneState] = rays / c->stats.ms_total;
            if (++c->tuneState == 4) {
                // trial 0 ran the default of a cache-resident scene, trial 1 that of a large one: another combination
                // replaces the scene's default only if it is clearly (1.5 %) faster
                const int defTrial = bigScene ? 1 : 0;
                int best = defTrial;
                for (int t = 0; t < 4; t++) if (c->tuneRate[t] > c->tuneRate[best]) best = t;
                if (c->tuneRate[best] <= 1.015 * c->tuneRate[defTrial]) best = defTrial;
                c->tuneChoice = kTuneTrial[best];
                c->tuneRateBase = c->tuneRate[best];      // (the pass-shape candidate is held against this; cached once that is decided)
                if (shape_user(c) || c->passStreams < 2) c->shapeState = 2;
            }
        }
    }
    if (shapeTrial) {
        const double rays = (double)(hs.rays_primary + hs.rays_bounce + hs.rays_shadow);
        if (rays >= 3.0e7 && c->stats.ms_total > 0.0 && c->allocations == allocationsBefore) {
            c->shapeChoice = rays / c->stats.ms_total > 1.02 * c->tuneRateBase ? 1 : 0;
            c->shapeState = 2;
        }
    }
    if ((trial || shapeTrial || shapeDecidedNow) && c->tuneState >= 4 && c->shapeState == 2 && c->tuneCache != 0) {
        TuneCache& T = tune_cache();
        std::lock_guard<std::mutex> g(T.mu);
        T.choice[c->sceneKey] = c->tuneChoice | (c->shapeChoice << 2);
    }
    if (count && std::getenv("RT_DEBUG"))
        std::fprintf(stderr, "[rt] refill census: node-phase exits %llu, parked lanes/exit %.2f, free lanes/exit %.2f; reasons: parked>=triMin %llu, few nodes %llu, refill %llu, no nodes %llu; deferred %llu\n",
                     hs.dbg[0], hs.dbg[0] ? (double)hs.dbg[1] / hs.dbg[0] : 0.0, hs.dbg[0] ? (double)hs.dbg[7] / hs.dbg[0] : 0.0, hs.dbg[2], hs.dbg[3], hs.dbg[5], hs.dbg[6], hs.dbg[4]);
    if (std::getenv("RT_DEBUG")) {
        std::vector<uint32_t> qc(2 * (maxB + 1));
        if (hipMemcpy(qc.data(), Pset[0].counters, qc.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
            std::fprintf(stderr, "[rt] queue sizes of the last pass on set 0 (paths / shadow rays per bounce), max ray steps %llu:", hs.max_ray_steps);
            for (uint32_t b = 0; b < maxB; b++) std::fprintf(stderr, " %u/%u", qc[b], qc[maxB + 1 + b]);
            std::fprintf(stderr, "\n");
        }
    }
    if (count && std::getenv("RT_DEBUG"))
        std::fprintf(stderr, "[rt] steps with no refill possible: node %llu (lanes/step %.1f) of %llu, tri %llu (lanes/step %.1f) of %llu\n",
                     hs.dbg2[0], hs.dbg2[0] ? (double)hs.dbg2[1] / hs.dbg2[0] : 0.0, hs.wave_node_steps, hs.dbg2[2], hs.dbg2[2] ? (double)hs.dbg2[3] / hs.dbg2[2] : 0.0, hs.wave_tri_steps);
    c->stats.shadow_node_visits = hs.shadow_node_visits; c->stats.shadow_tri_tests = hs.shadow_tri_tests;
    c->stats.max_ray_steps = hs.max_ray_steps;
    c->stats.bounce_node_visits = hs.bounce_node_visits; c->stats.bounce_tri_tests = hs.bounce_tri_tests;
    c->stats.wave_node_steps = hs.wave_node_steps; c->stats.wave_tri_steps = hs.wave_tri_steps;
    c->stats.ms_extend = msAcc[0] + msAcc[3]; c->stats.ms_shade = msAcc[1]; c->stats.ms_shadow = msAcc[2];
    c->stats.launches_extend = nLaunch[0] + nLaunch[3]; c->stats.launches_shade = nLaunch[1]; c->stats.launches_shadow = nLaunch[2];
    c->stats.ms_extend_primary = msAcc[3]; c->stats.launches_extend_primary = nLaunch[3];
    c->stats.launches_pair = nPairs;
    c->stats.wide_nodes = (wide && (refillExtend || refillShadow)) ? 1 : 0;
    c->stats.wide_depth = (int32_t)c->cs.wideDepth;
    { const LogShape g = log_shape(c, nslots, maxB); c->stats.log_dense_bounces = (int32_t)(g.lateBounces ? g.dense : maxB); c->stats.log_fallbacks = (int32_t)c->lateLogFallbacks; }
    // the combination the NEXT frame will run: on the frame that finishes the trial `combo` is still the last trial's
    // (bit 2: the pass shape, as the next frame will have it)
    const int shapeNext = shape_user(c) || c->passStreams < 2 ? 0 : c->kernelChoice >= 0 ? ((c->kernelChoice >> 2) & 1) : c->shapeState == 2 ? c->shapeChoice : 0;
    c->stats.kernel_choice = exact ? -2 : (((trial && c->tuneState >= 4) ? c->tuneChoice : combo) | (shapeNext << 2));
    c->stats.kernel_choice_settled = (exact || c->refill >= 0 || c->kernelChoice >= 0 || (c->tuneState >= 4 && (c->shapeState == 2 || shape_user(c) || c->passStreams < 2))) ? 1 : 0;
    c->primValid = aov;
    return RT_OK;
}

// rt_trace_rays_noalloc: what rt_trace_rays computes on the host from the hit records, on the device, so that the results
// go straight from `scratch` into the caller's arrays: reference triangle id and hitLoc = orig + dir * t
// (shader.comp:122; -ffp-contract=off, the same two roundings as the host loop of rt_trace_rays).
__global__ void k_hits_out(const float2* __restrict__ hit, const float* __restrict__ origins, const float* __restrict__ dirs,
                           uint32_t n, int32_t* __restrict__ triOut, float* __restrict__ locOut) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 h = hit[i];
    const int32_t id = __float_as_int(h.y);
    triOut[i] = id;
    for (int k = 0; k < 3; k++) {
        const float prod = dirs[3 * i + k] * h.x;
        locOut[3 * i + k] = id >= 0 ? origins[3 * i + k] + prod : 0.f;
    }
}

// RGBA32F image <-> tightly packed RGB (OIDN's Float3 layout; main.cpp:157-170 rgbaTOrgb and 176-183 on the way back)
__global__ void k_pack_rgb(const float4* __restrict__ img, float* __restrict__ rgb, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = img[i];
    rgb[3 * i] = v.x; rgb[3 * i + 1] = v.y; rgb[3 * i + 2] = v.z;
}
__global__ void k_unpack_rgb(float4* __restrict__ img, const float* __restrict__ rgb, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    img[i] = make_float4(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], 1.f);
}

// Renderer.cpp:103-107 + the vertical flip of 121-127: clamp to [0, 1], * 255, truncating cast; top row first.
// (A NaN passes std::clamp and converts to 0 on the reference's x86 build; here it is written as 0 explicitly.)
__global__ void k_frame_rgba8(const float4* __restrict__ img, uchar4* __restrict__ out, uint32_t w, uint32_t h) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    const uint32_t y = i / w, x = i - y * w;
    const float4 v = img[(size_t)(h - 1u - y) * w + x];
    auto cv = [](float f) -> unsigned char {
        f = f < 0.f ? 0.f : (f > 1.f ? 1.f : f);
        return f == f ? static_cast<unsigned char>(f * 255.f) : (unsigned char)0;
    };
    out[i] = make_uchar4(cv(v.x), cv(v.y), cv(v.z), cv(v.w));
}

__global__ void k_add_image(float4* __restrict__ dst, const float4* __restrict__ src, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = dst[i]; const float4 b = src[i];
    a.x = a.x + b.x; a.y = a.y + b.y; a.z = a.z + b.z; a.w = a.w + b.w;
    dst[i] = a;
}

bool name_is(const char* a, const char* b) { return std::strcmp(a, b) == 0; }

}  // namespace

extern "C" {

rt_status rt_create(rt_ctx** out, const rt_config* cfg) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int dev = cfg ? cfg->device : 0;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device available (librt_core has no CPU fallback)");
    if (dev < 0 || dev >= ndev) return fail(nullptr, RT_ERR_NO_DEVICE, "device ordinal out of range");
    if (hipSetDevice(dev) != hipSuccess) return fail(nullptr, RT_ERR_NO_DEVICE, "hipSetDevice failed");
    if (dev < kMaxDevices) {
        // this device's copy of the file-scope scratch array (rt_trace_rays_noalloc), looked up once, with the device current
        // (every caller has just made `dev` current on its own thread, so whichever thread runs the lookup has too)
        std::call_once(g_scratchOnce[dev], [dev] {
            void* sp = nullptr;
            g_scratchErr[dev] = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));
            if (g_scratchErr[dev] == hipSuccess) g_scratchPtr[dev] = (unsigned char*)sp;
        });
        if (!g_scratchPtr[dev]) return fail(nullptr, RT_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(g_scratchErr[dev]));
    }
    rt_ctx* c = new rt_ctx();
    c->device = dev;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) { c->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; if (prop.totalGlobalMem) c->deviceBytes = prop.totalGlobalMem; }
    bool ok = hipStreamCreate(&c->stream) == hipSuccess && hipEventCreateWithFlags(&c->evInit, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; ok && k < rt_ctx::kMaxSets; k++) {
        ok = hipEventCreateWithFlags(&c->evResolve[k], hipEventDisableTiming) == hipSuccess;
        if (ok && k > 0) ok = hipStreamCreate(&c->extra[k]) == hipSuccess;
    }
    if (!ok) {
        rt_destroy(c);
        return fail(nullptr, RT_ERR_HIP, "stream / event creation failed");
    }
    *out = c;
    return RT_OK;
}

void rt_destroy(rt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (hipStream_t x : c->extra) if (x) (void)hipStreamSynchronize(x);
    release_paths(c);
    for (hipEvent_t ev : c->evResolve) if (ev) (void)hipEventDestroy(ev);
    if (c->evInit) (void)hipEventDestroy(c->evInit);
    for (hipStream_t x : c->extra) if (x) (void)hipStreamDestroy(x);
    DevBuf* bufs[] = {&c->dNodes, &c->dWNodes, &c->dTris, &c->dShade, &c->dShadeUV, &c->dCMat, &c->dMaterials, &c->dVertices, &c->dLights, &c->dTexDescs,
                      &c->img[0], &c->img[1], &c->img[2], &c->img[3], &c->primTri, &c->primHit, &c->accum, &c->dStats, &c->rgbStage};
    for (DevBuf* b : bufs) b->release();
    for (auto& t : c->textures) t.data.release();
    for (auto& e : c->evRing) { if (e.a) (void)hipEventDestroy(e.a); if (e.b) (void)hipEventDestroy(e.b); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* rt_last_error(const rt_ctx* c) { return c ? c->err.c_str() : g_createErr.c_str(); }

rt_status rt_upload_buffer(rt_ctx* c, uint32_t slot, const void* data, uint64_t bytes) {
    if (!c) return RT_ERR_INVALID;
    if (slot > 5) return fail(c, RT_ERR_INVALID, "slot must be 0..5");
    if (bytes && !data) return fail(c, RT_ERR_INVALID, "data is NULL");
    if (bytes % kRecordSize[slot]) return fail(c, RT_ERR_INVALID, "size is not a multiple of the slot's record size");
    try { c->slotHost[slot].assign((const uint8_t*)data, (size_t)bytes); }
    catch (const std::bad_alloc&) { return fail(c, RT_ERR_HIP, "out of host memory for the copy of slot " + std::to_string(slot)); }
    c->slotSet[slot] = true;
    if (slot == RT_SLOT_LIGHTS) c->lightsDirty = true;
    else if (slot == RT_SLOT_MATERIALS) { c->materialsDirty = true; c->sceneDirty = true; }
    else c->sceneDirty = true;
    return RT_OK;
}

rt_status rt_set_int(rt_ctx* c, const char* name, int32_t v) {
    if (!c || !name) return RT_ERR_INVALID;
    if (name_is(name, "numLights")) c->numLights = v;
    else if (name_is(name, "hasSun")) c->hasSun = v;
    else if (name_is(name, "samplesPerPixel")) c->samplesPerPixel = v;
    else if (name_is(name, "maxBounces")) c->maxBounces = v;
    else if (name_is(name, "traversalMode")) c->traversalMode = v;
    else if (name_is(name, "timeKernels")) c->timeKernels = v;
    else if (name_is(name, "countVisits")) c->countVisits = v;
    else if (name_is(name, "samplesPerPass")) { c->samplesPerPass = v < 1 ? 64 : v; c->sppUser = v >= 1 && v != 64; }      // < 1: back to the default
    else if (name_is(name, "sampleGroup")) c->sampleGroup = v;
    else if (name_is(name, "pathMemPercent")) c->pathMemPercent = v;
    else if (name_is(name, "logDenseBounces")) { c->logDenseBounces = v; c->lateLogOff = false; }
    else if (name_is(name, "lateLogPercent")) { c->lateLogPercent = v; c->lateLogOff = false; }
    else if (name_is(name, "blocksPerCU")) c->blocksPerCU = v < 1 ? 1 : v;
    else if (name_is(name, "refill")) c->refill = v;
    else if (name_is(name, "refillIdle")) c->refillIdle = v;
    else if (name_is(name, "refillFromBounce")) c->refillFromBounce = v < 1 ? 1 : v;
    else if (name_is(name, "packetBounces")) c->packetBounces = v < 0 ? -1 : v;
    else if (name_is(name, "refillNodeMin")) c->refillNodeMin = v;
    else if (name_is(name, "refillTriMin")) c->refillTriMin = v;
    else if (name_is(name, "refillNodeLow")) c->refillNodeLow = v;
    else if (name_is(name, "refillDefer")) c->refillDefer = v;
    else if (name_is(name, "wideNodes")) c->wideNodes = v;
    else if (name_is(name, "sunKernel")) c->sunKernel = v;
    else if (name_is(name, "tuneCache")) c->tuneCache = v;
    else if (name_is(name, "kernelChoice")) c->kernelChoice = v < 0 ? -1 : (v & 7);      // bits 0, 1: kernels; bit 2: pass shape
    else if (name_is(name, "refillBlocksPerCU")) c->refillBlocksPerCU = v < 0 ? 0 : v;
    else if (name_is(name, "autoPass")) c->autoPass = v;
    else if (name_is(name, "fusePairs")) c->fusePairs = v;
    else if (name_is(name, "passStreams")) { c->passStreams = v < 1 ? 4 : v; c->streamsUser = v >= 1; }      // < 1: back to the default (and to the measured pass shape)
    return RT_OK;   // unknown names are ignored (glGetUniformLocation == -1)
}
rt_status rt_set_float(rt_ctx* c, const char* name, float v) {
    if (!c || !name) return RT_ERR_INVALID;
    if (name_is(name, "aspect")) c->aspect = v;
    else if (name_is(name, "fov")) c->fov = v;
    return RT_OK;
}
rt_status rt_set_vec2(rt_ctx* c, const char* name, const float v[2]) {
    if (!c || !name || !v) return RT_ERR_INVALID;
    if (name_is(name, "offset")) { c->offset[0] = v[0]; c->offset[1] = v[1]; }   // tile origin: scheduling only (main.cpp:126-139)
    return RT_OK;
}
rt_status rt_set_vec3(rt_ctx* c, const char* name, const float v[3]) {
    if (!c || !name || !v) return RT_ERR_INVALID;
    float* dst = nullptr;
    if (name_is(name, "camPos")) dst = c->camPos;
    else if (name_is(name, "camForward")) dst = c->camForward;
    else if (name_is(name, "camRight")) dst = c->camRight;
    else if (name_is(name, "camUp")) dst = c->camUp;
    if (dst) { dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; }
    return RT_OK;
}
rt_status rt_set_vec4(rt_ctx* c, const char* name, const float v[4]) { return (!c || !name || !v) ? RT_ERR_INVALID : RT_OK; }
rt_status rt_set_mat4(rt_ctx* c, const char* name, const float m[16]) { return (!c || !name || !m) ? RT_ERR_INVALID : RT_OK; }

rt_status rt_resize(rt_ctx* c, uint32_t w, uint32_t h) {
    if (!c) return RT_ERR_INVALID;
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 28)) return fail(c, RT_ERR_INVALID, "bad image size");
    HIPCHK(c, hipSetDevice(c->device));
    size_t n = (size_t)w * h;
    rt_status s;
    for (int i = 0; i < 4; i++) {
        if ((s = ensure(c, c->img[i], n * 16))) return s;
        HIPCHK(c, hipMemsetAsync(c->img[i].p, 0, n * 16, c->stream));
    }
    if ((s = ensure(c, c->primTri, n * 4))) return s;
    if ((s = ensure(c, c->primHit, n * 12))) return s;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->width = w; c->height = h; c->primValid = false;
    return RT_OK;
}

rt_status rt_create_texture(rt_ctx* c, const uint8_t* rgba8, uint32_t w, uint32_t h, uint64_t* handle_out) {
    if (!c || !rgba8 || !handle_out || w == 0 || h == 0) return c ? fail(c, RT_ERR_INVALID, "bad texture arguments") : RT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    Texture t; t.w = w; t.h = h;
    rt_status s = upload(c, t.data, rgba8, (size_t)w * h * 4);
    if (s) return s;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->textures.push_back(t);
    c->texturesDirty = true; c->materialsDirty = true;
    *handle_out = (uint64_t)c->textures.size();
    return RT_OK;
}

rt_status rt_render_frame(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    return render_region(c, 0, 0, c->width, c->height, 0, (uint32_t)(c->samplesPerPixel > 0 ? c->samplesPerPixel : 0));
}
rt_status rt_render_region(rt_ctx* c, uint32_t x, uint32_t y, uint32_t w, uint32_t h, uint32_t r0, uint32_t r1) {
    return render_region(c, x, y, w, h, r0, r1);
}

rt_status rt_read_image(rt_ctx* c, uint32_t image, float* dst) {
    if (!c) return RT_ERR_INVALID;
    if (image > 3 || !dst) return fail(c, RT_ERR_INVALID, "image must be 0..3 and dst non-NULL");
    if (!c->img[image].p) return fail(c, RT_ERR_NOT_READY, "rt_resize has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(dst, c->img[image].p, (size_t)c->width * c->height * 16, hipMemcpyDeviceToHost));
    return RT_OK;
}
rt_status rt_write_image(rt_ctx* c, uint32_t image, const float* src) {
    if (!c) return RT_ERR_INVALID;
    if (image > 3 || !src) return fail(c, RT_ERR_INVALID, "image must be 0..3 and src non-NULL");
    if (!c->img[image].p) return fail(c, RT_ERR_NOT_READY, "rt_resize has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(c->img[image].p, src, (size_t)c->width * c->height * 16, hipMemcpyHostToDevice));
    return RT_OK;
}
rt_status rt_read_image_rgb(rt_ctx* c, uint32_t image, float* dst) {
    if (!c) return RT_ERR_INVALID;
    if (image > 3 || !dst) return fail(c, RT_ERR_INVALID, "image must be 0..3 and dst non-NULL");
    if (!c->img[image].p) return fail(c, RT_ERR_NOT_READY, "rt_resize has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t n = c->width * c->height;
    rt_status s = ensure(c, c->rgbStage, (size_t)n * 12);
    if (s) return s;
    k_pack_rgb<<<(n + 255) / 256, 256, 0, c->stream>>>((const float4*)c->img[image].p, (float*)c->rgbStage.p, n);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(dst, c->rgbStage.p, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}
rt_status rt_write_image_rgb(rt_ctx* c, uint32_t image, const float* src) {
    if (!c) return RT_ERR_INVALID;
    if (image > 3 || !src) return fail(c, RT_ERR_INVALID, "image must be 0..3 and src non-NULL");
    if (!c->img[image].p) return fail(c, RT_ERR_NOT_READY, "rt_resize has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t n = c->width * c->height;
    rt_status s = ensure(c, c->rgbStage, (size_t)n * 12);
    if (s) return s;
    HIPCHK(c, hipMemcpyAsync(c->rgbStage.p, src, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
    k_unpack_rgb<<<(n + 255) / 256, 256, 0, c->stream>>>((float4*)c->img[image].p, (const float*)c->rgbStage.p, n);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}
rt_status rt_read_frame_rgba8(rt_ctx* c, uint8_t* dst, uint64_t dst_bytes) {
    if (!c) return RT_ERR_INVALID;
    if (!dst) return fail(c, RT_ERR_INVALID, "dst must be non-NULL");
    if (!c->img[0].p) return fail(c, RT_ERR_NOT_READY, "rt_resize has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t n = c->width * c->height;
    if (dst_bytes != (uint64_t)n * 4) return fail(c, RT_ERR_INVALID, "dst_bytes must be width * height * 4 of the current image size");
    rt_status s = ensure(c, c->rgbStage, (size_t)n * 12);          // shared staging area (n * 4 bytes used here)
    if (s) return s;
    k_frame_rgba8<<<(n + 255) / 256, 256, 0, c->stream>>>((const float4*)c->img[0].p, (uchar4*)c->rgbStage.p, c->width, c->height);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(dst, c->rgbStage.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}
rt_status rt_image_device_ptr(rt_ctx* c, uint32_t image, void** p) {
    if (!c) return RT_ERR_INVALID;
    if (image > 3 || !p) return fail(c, RT_ERR_INVALID, "image must be 0..3");
    if (!c->img[image].p) return fail(c, RT_ERR_NOT_READY, "rt_resize has not been called");
    *p = c->img[image].p;
    return RT_OK;
}

rt_status rt_add_image(rt_ctx* c, uint32_t image, const void* src) {
    if (!c) return RT_ERR_INVALID;
    if (image > 3 || !src) return fail(c, RT_ERR_INVALID, "image must be 0..3 and src non-NULL");
    if (!c->img[image].p) return fail(c, RT_ERR_NOT_READY, "rt_resize has not been called");
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, src) != hipSuccess || at.device != c->device) {
        (void)hipGetLastError();
        return fail(c, RT_ERR_INVALID, "src is not device memory of this context's GPU");
    }
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t n = c->width * c->height;
    k_add_image<<<(n + 255) / 256, 256, 0, c->stream>>>((float4*)c->img[image].p, (const float4*)src, n);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

rt_status rt_get_stats(rt_ctx* c, rt_stats* out) {
    if (!c || !out) return RT_ERR_INVALID;
    *out = c->stats;
    return RT_OK;
}

rt_status rt_read_primary_hits(rt_ctx* c, int32_t* tri, float* hits) {
    if (!c) return RT_ERR_INVALID;
    if (!c->primValid) return fail(c, RT_ERR_NOT_READY, "no render containing sample 0 has run");
    HIPCHK(c, hipSetDevice(c->device));
    size_t n = (size_t)c->width * c->height;
    if (tri) HIPCHK(c, hipMemcpy(tri, c->primTri.p, n * 4, hipMemcpyDeviceToHost));
    if (hits) HIPCHK(c, hipMemcpy(hits, c->primHit.p, n * 12, hipMemcpyDeviceToHost));
    return RT_OK;
}

rt_status rt_trace_rays(rt_ctx* c, const float* origins, const float* dirs, uint64_t n, int32_t* tri, float* hits) {
    if (!c) return RT_ERR_INVALID;
    if (!origins || !dirs || !tri) return fail(c, RT_ERR_INVALID, "NULL argument");
    if (n == 0) return RT_OK;
    if (n > (1ull << 28)) return fail(c, RT_ERR_INVALID, "too many rays in one call");
    // lights are not needed for a closest-hit query
    bool hadLights = c->slotSet[RT_SLOT_LIGHTS];
    if (!hadLights) { c->slotSet[RT_SLOT_LIGHTS] = true; c->lightsDirty = true; }
    rt_status s = sync_scene(c);
    if (!hadLights) c->slotSet[RT_SLOT_LIGHTS] = false;
    if (s) return s;
    uint32_t nslots = (uint32_t)((n + 63) / 64 * 64);
    if ((s = ensure_paths(c, 1, nslots, nslots, 1, 1))) return s;
    PathBuffers P = make_paths(c, 0, nslots, nslots, 1, 1);
    const bool exact = use_exact(c);
    const bool count = c->countVisits != 0;
    SceneView S = make_view(c, exact);
    DevBuf dO, dD;
    if ((s = upload(c, dO, origins, n * 12))) return s;
    if ((s = upload(c, dD, dirs, n * 12))) { dO.release(); return s; }
    hipStream_t st = c->stream;
    DeviceStats* dst = (DeviceStats*)c->dStats.p;
    (void)hipMemsetAsync(dst, 0, sizeof(DeviceStats), st);
    uint32_t blocks = (nslots + RT_BLOCK - 1) / RT_BLOCK;
    k_load_rays<<<blocks, RT_BLOCK, 0, st>>>(P, (const float*)dO.p, (const float*)dD.p, (uint32_t)n, P.counters);
    FrameParams F0;
    std::memset(&F0, 0, sizeof F0);
    launch_extend<false>(exact, count, dim3(trace_blocks(c, nslots)), st, S, F0, P, (const float4*)P.ray[0],
                             (const uint32_t*)P.counters, 0, 0u, dst);
    hipError_t e = hipStreamSynchronize(st);
    std::vector<float> h2(n * 2);
    if (e == hipSuccess) e = hipMemcpy(h2.data(), P.hit, n * 8, hipMemcpyDeviceToHost);
    DeviceStats hs; std::memset(&hs, 0, sizeof hs);
    if (e == hipSuccess) e = hipMemcpy(&hs, dst, sizeof hs, hipMemcpyDeviceToHost);
    dO.release(); dD.release();
    if (e != hipSuccess) return fail(c, RT_ERR_HIP, std::string("rt_trace_rays: ") + hipGetErrorString(e));
    for (uint64_t i = 0; i < n; i++) {
        int32_t id; std::memcpy(&id, &h2[i * 2 + 1], 4);
        tri[i] = id;
        if (hits) {
            // hitLoc = orig + dir * t (shader.comp:122), the same float32 operations the kernels use (-ffp-contract=off)
            const float t = h2[i * 2];
            for (int k = 0; k < 3; k++) {
                volatile float prod = dirs[i * 3 + k] * t;
                hits[i * 3 + k] = id >= 0 ? origins[i * 3 + k] + prod : 0.f;
            }
        }
    }
    std::memset(&c->stats, 0, sizeof c->stats);
    c->stats.rays_primary = hs.rays_primary + hs.rays_bounce; c->stats.node_visits = hs.node_visits; c->stats.tri_tests = hs.tri_tests;
    return RT_OK;
}

rt_status rt_trace_rays_noalloc(rt_ctx* c, const float* origins, const float* dirs, uint64_t n, int32_t* tri, float* hits) {
    if (!c) return RT_ERR_INVALID;
    if (!origins || !dirs || !tri || !hits) return fail(c, RT_ERR_INVALID, "NULL argument");
    if (n == 0) return RT_OK;
    if (n > kNoAllocMaxRays) return fail(c, RT_ERR_INVALID, "rt_trace_rays_noalloc takes at most RT_NOALLOC_MAX_RAYS (" + std::to_string(kNoAllocMaxRays) + ") rays per call");
    // Deriving and uploading the compact scene allocates: it must have happened already (any rt_render_* or rt_trace_rays
    // call after the last upload of slots 0..4 does it).
    for (int s : {RT_SLOT_VERTICES, RT_SLOT_MESH_MATRICES, RT_SLOT_MATERIALS, RT_SLOT_BVH_NODES, RT_SLOT_TRIANGLES})
        if (!c->slotSet[s]) return fail(c, RT_ERR_NOT_READY, "slot " + std::to_string(s) + " has not been uploaded");
    if (c->sceneDirty || !c->dNodes.p || !c->dTris.p)
        return fail(c, RT_ERR_NOT_READY, "the scene is not resident yet: call rt_render_frame / rt_render_region / rt_trace_rays once after uploading it");
    if (c->device >= kMaxDevices || !g_scratchPtr[c->device]) return fail(c, RT_ERR_HIP, "no scratch address for this device");
    HIPCHK(c, hipSetDevice(c->device));
    unsigned char* base = g_scratchPtr[c->device];      // never the array's name: that is not a device address here
    const uint32_t nn = (uint32_t)n;
    const uint32_t nslots = (nn + 63u) / 64u * 64u;
    const uint32_t blocks = (nslots + RT_BLOCK - 1) / RT_BLOCK;
    auto up = [](uint64_t v) { return (v + 255) / 256 * 256; };
    uint64_t off = kScratchHeader;
    float4* dRay = (float4*)(base + off);   off = up(off + 32ull * nslots);
    float2* dHit = (float2*)(base + off);   off = up(off + 8ull * nslots);
    float* dO = (float*)(base + off);       off = up(off + 12ull * nn);
    float* dD = (float*)(base + off);       off = up(off + 12ull * nn);
    int32_t* dTri = (int32_t*)(base + off); off = up(off + 4ull * nn);
    float* dLoc = (float*)(base + off);     off = up(off + 12ull * nn);
    uint2* dOv = (uint2*)(base + off);
    const uint32_t ovDepth = overflow_depth(c), ovStride = blocks * RT_BLOCK;
    off += 8ull * ovDepth * ovStride;
    if (off > kScratchBytes) return fail(c, RT_ERR_INVALID, "internal error: scratch layout exceeds the array");
    uint32_t* dCount = (uint32_t*)base;
    DeviceStats* dst = (DeviceStats*)(base + 64);
    static_assert(64 + sizeof(DeviceStats) <= kScratchHeader, "scratch header");

    PathBuffers P;
    std::memset(&P, 0, sizeof P);
    P.ray[0] = dRay; P.hit = dHit; P.counters = dCount;
    P.nslots = nslots; P.slotsPerSample = nslots; P.lightSlots = 1; P.grpShift = 0;
    P.late = nullptr; P.lateNext = nullptr; P.lateHead = nullptr; P.denseBounces = 0xffffffffu; P.lateCap = 0;
    P.overflow = dOv; P.overflowStride = ovStride; P.overflowDepth = ovDepth;
    const bool exact = use_exact(c);
    const bool count = c->countVisits != 0;
    SceneView S = make_view(c, exact);
    hipStream_t st = c->stream;
    // the array keeps the previous call's contents: the header (queue counter, statistics) starts at zero; everything else
    // a call reads it has written first (rays, hits and results for i < n; stack entries are pushed before they are popped)
    // From here on work is in flight on the shared array: an error return drains the stream first, so that the next call on
    // this GPU cannot overlap what this one left running.
    DeviceStats hs;
    hipError_t e = hipMemsetAsync(base, 0, kScratchHeader, st);
    const char* what = "hipMemsetAsync(header)";
    auto step = [&](hipError_t r, const char* w) { if (e == hipSuccess && r != hipSuccess) { e = r; what = w; } return e == hipSuccess; };
    if (e == hipSuccess) step(hipMemcpyAsync(dO, origins, 12ull * nn, hipMemcpyHostToDevice, st), "copy of origins");
    if (e == hipSuccess) step(hipMemcpyAsync(dD, dirs, 12ull * nn, hipMemcpyHostToDevice, st), "copy of dirs");
    if (e == hipSuccess) {
        k_load_rays<<<blocks, RT_BLOCK, 0, st>>>(P, dO, dD, nn, dCount);
        FrameParams F0;
        std::memset(&F0, 0, sizeof F0);
        launch_extend<false>(exact, count, dim3(blocks), st, S, F0, P, (const float4*)dRay, (const uint32_t*)dCount, 0, 0u, dst);
        k_hits_out<<<(nn + 255u) / 256u, 256, 0, st>>>(dHit, dO, dD, nn, dTri, dLoc);
        step(hipGetLastError(), "kernel launch");
    }
    if (e == hipSuccess) step(hipMemcpyAsync(tri, dTri, 4ull * nn, hipMemcpyDeviceToHost, st), "copy of tri_ids");
    if (e == hipSuccess) step(hipMemcpyAsync(hits, dLoc, 12ull * nn, hipMemcpyDeviceToHost, st), "copy of hit_locs");
    if (e == hipSuccess) step(hipMemcpyAsync(&hs, dst, sizeof hs, hipMemcpyDeviceToHost, st), "copy of statistics");
    {
        const hipError_t se = hipStreamSynchronize(st);      // always: the call returns with nothing of its own in flight
        if (e == hipSuccess && se != hipSuccess) { e = se; what = "hipStreamSynchronize"; }
    }
    if (e != hipSuccess) return fail(c, RT_ERR_HIP, std::string("rt_trace_rays_noalloc: ") + what + ": " + hipGetErrorString(e));
    std::memset(&c->stats, 0, sizeof c->stats);
    c->stats.rays_primary = hs.rays_primary + hs.rays_bounce; c->stats.node_visits = hs.node_visits; c->stats.tri_tests = hs.tri_tests;
    return RT_OK;
}

}  // extern "C"
