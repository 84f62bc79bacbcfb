// The multi-GPU frame driver of rt_host.hpp (MultiGPU) and its C entries (include/rt_host.h, rth_multi_*).  Kept apart
// from rt_host.cpp because it is the one part of the host layer that talks to the HIP runtime itself (device selection,
// the reduce's streams) and to RCCL: rt_host.cpp stays plain C++ over the C ABI, which is what tools/sanitize builds.
#include "rt_host.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "../../../include/rt_host.h"

namespace rt_host {

void set_last_error(const std::string& m);      // rt_host.cpp: the message rth_last_error returns on this thread

static void chk(rt_ctx* c, rt_status s, const char* what) {
    if (s != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error(c));
}

// ---- Multi-GPU frame driver ----------------------------------------------------------------------------
// Partition: the index math of raytracing_amd/distributed.py (sample_range, row_band, partition), tested against it.
static void sample_range(uint32_t spp, int rank, int world, uint32_t& r0, uint32_t& r1) {
    const uint32_t base = spp / (uint32_t)world, extra = spp % (uint32_t)world;
    r0 = (uint32_t)rank * base + std::min<uint32_t>((uint32_t)rank, extra);
    r1 = r0 + base + ((uint32_t)rank < extra ? 1u : 0u);
}
static void row_band(uint32_t height, int rank, int world, uint32_t& y0, uint32_t& y1) {
    const uint32_t align = 8, tiles = (height + align - 1) / align;
    const uint32_t base = tiles / (uint32_t)world, extra = tiles % (uint32_t)world;
    const uint32_t t0 = (uint32_t)rank * base + std::min<uint32_t>((uint32_t)rank, extra);
    const uint32_t t1 = t0 + base + ((uint32_t)rank < extra ? 1u : 0u);
    y0 = std::min(t0 * align, height); y1 = std::min(t1 * align, height);
}
Shard partition(uint32_t width, uint32_t height, uint32_t spp, int rank, int world, int bandRanks) {
    if (world < 1 || rank < 0 || rank >= world || bandRanks < 1 || world % bandRanks != 0) throw std::invalid_argument("partition: bad rank / world / band_ranks");
    const int sampleRanks = world / bandRanks;
    const int b = rank / sampleRanks, sidx = rank % sampleRanks;
    uint32_t y0, y1;
    Shard s;
    row_band(height, b, bandRanks, y0, y1);
    sample_range(spp, sidx, sampleRanks, s.r0, s.r1);
    s.x = 0; s.y = y0; s.w = width; s.h = y1 - y0;
    return s;
}

// librccl.so.1, loaded on first use: a job on one GPU (and the library itself) carries no dependency on it.
namespace {
struct Rccl {
    typedef void* comm_t;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    static constexpr int kFloat = 7, kSum = 0;      // ncclFloat32, ncclSum (rccl.h)
    static Rccl& get() {
        static Rccl r;
        static std::once_flag once;
        std::call_once(once, [] {
            void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) return;
            r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
            r.Reduce = (decltype(r.Reduce))dlsym(h, "ncclReduce");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
        });
        return r;
    }
    bool ok() const { return CommInitAll && CommDestroy && Reduce; }
};
}  // namespace

struct MultiGPU::Impl {
    // Contexts are grouped by physical GPU: `leader[g]` is the first context on distinct GPU g; images of the other contexts on
    // that GPU are added onto the leader's (rt_add_image), then the leaders reduce onto leader[0] == context 0 through RCCL.
    std::vector<size_t> leader;
    std::vector<size_t> groupOf;                 // context -> index into leader
    std::vector<Rccl::comm_t> comms;             // one per distinct GPU, when there are several
    std::vector<hipStream_t> streams;            // the reduce's stream on each distinct GPU
};

MultiGPU::MultiGPU(const std::vector<int>& devs) : devices(devs) {
    if (devs.empty()) throw std::invalid_argument("MultiGPU: no devices");
    impl = new Impl();
    try {
        for (int d : devs) shaders.push_back(new ComputeShader("shader.comp", d));
        impl->groupOf.resize(devs.size());
        for (size_t i = 0; i < devs.size(); i++) {
            size_t g = 0;
            while (g < impl->leader.size() && devices[impl->leader[g]] != devs[i]) g++;
            if (g == impl->leader.size()) impl->leader.push_back(i);
            impl->groupOf[i] = g;
        }
    } catch (...) {
        for (ComputeShader* s : shaders) delete s;
        delete impl;
        throw;
    }
}

MultiGPU::~MultiGPU() {
    for (size_t g = 0; g < impl->comms.size(); g++) {
        (void)hipSetDevice(devices[impl->leader[g]]);
        if (impl->comms[g]) (void)Rccl::get().CommDestroy(impl->comms[g]);
    }
    for (size_t g = 0; g < impl->streams.size(); g++) {
        (void)hipSetDevice(devices[impl->leader[g]]);
        if (impl->streams[g]) (void)hipStreamDestroy(impl->streams[g]);
    }
    for (ComputeShader* s : shaders) delete s;
    delete impl;
}

void MultiGPU::ReplicateScene() {
    for (int slot = 0; slot < 6; slot++)
        if (SSBO::last[slot].size() > 0xffffffffull) throw std::length_error("MultiGPU::ReplicateScene: a scene buffer exceeds SSBO::Bind's 32-bit size");
    int numLights = (int)(SSBO::last[5].size() / sizeof(GPULight));
    bool hasSun = false;
    for (int l = 0; l < numLights; l++) if (((const GPULight*)SSBO::last[5].data())[l].lightType == 1u) hasSun = true;
    // (copies first: SSBO::Bind stores what it is handed into SSBO::last itself)
    std::vector<uint8_t> bufs[6];
    for (int slot = 0; slot < 6; slot++) bufs[slot] = SSBO::last[slot];
    for (size_t i = 1; i < shaders.size(); i++) {
        shaders[i]->Activate();
        for (size_t t = 0; t < Model::textureLog.size(); t++) {
            const ImageRGBA8& im = Model::textureLog[t];
            uint64_t handle = 0;
            chk(shaders[i]->ID, rt_create_texture(shaders[i]->ID, im.rgba.data(), im.width, im.height, &handle), "rt_create_texture (replica)");
            if (handle != t + 1) throw std::runtime_error("MultiGPU::ReplicateScene: a replica's texture handles differ from context 0's (it already held textures)");
        }
        for (int slot = 0; slot < 6; slot++) SSBO::Bind(bufs[slot].data(), (uint32_t)bufs[slot].size(), (uint32_t)slot);
        shaders[i]->setInt("numLights", numLights);
        shaders[i]->setInt("hasSun", hasSun);
    }
    shaders[0]->Activate();
    framesRendered = 0; kernelsAgreed = false;
}

void MultiGPU::setInt(const std::string& key, int value) { for (ComputeShader* s : shaders) s->setInt(key, value); }
void MultiGPU::resize(uint32_t w, uint32_t h) {
    for (ComputeShader* s : shaders) chk(s->ID, rt_resize(s->ID, w, h), "rt_resize");
    width = w; height = h;
}
void MultiGPU::SetFrameUniforms(const Camera* camera, vec2 screenSize) {
    for (ComputeShader* s : shaders) rt_host::SetFrameUniforms(s, camera, screenSize);
    shaders[0]->Activate();
}

void MultiGPU::RenderFrame(uint32_t spp, int bandRanks, bool mergeAovs) {
    const int world = (int)shaders.size();
    if (width == 0 || height == 0) throw std::runtime_error("MultiGPU::RenderFrame: resize has not been called");
    setInt("samplesPerPixel", (int)spp);
    // one host thread per context (the reference's loop has one GL thread for its one GPU, main.cpp:95-141)
    std::vector<std::string> errs((size_t)world);
    auto work = [&](int rank) {
        try {
            rt_ctx* c = shaders[(size_t)rank]->ID;
            const Shard s = partition(width, height, spp, rank, world, bandRanks);
            // an empty shard still clears its images (a region of 8 rows, no samples: traces nothing)
            if (s.empty()) chk(c, rt_render_region(c, 0, 0, width, std::min<uint32_t>(8, height), 0, 0), "rt_render_region (empty shard)");
            else chk(c, rt_render_region(c, s.x, s.y, s.w, s.h, s.r0, s.r1), "rt_render_region");
        } catch (const std::exception& e) { errs[(size_t)rank] = e.what(); }
    };
    if (world == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int r = 0; r < world; r++) th.emplace_back(work, r);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < world; r++) if (!errs[(size_t)r].empty()) throw std::runtime_error("rank " + std::to_string(r) + ": " + errs[(size_t)r]);
    lastUsedRccl = false;
    if (world > 1) {
        const uint32_t nImages = mergeAovs ? 4u : 1u;
        // contexts that share a GPU: onto that GPU's first context
        for (size_t i = 0; i < shaders.size(); i++) {
            const size_t lead = impl->leader[impl->groupOf[i]];
            if (lead == i) continue;
            for (uint32_t img = 0; img < nImages; img++) {
                void* src = nullptr;
                chk(shaders[i]->ID, rt_image_device_ptr(shaders[i]->ID, img, &src), "rt_image_device_ptr");
                chk(shaders[lead]->ID, rt_add_image(shaders[lead]->ID, img, src), "rt_add_image");
            }
        }
        const size_t nGpus = impl->leader.size();
        if (nGpus > 1) {
            // the one collective: sum of the leaders' images onto context 0 (RCCL over xGMI)
            Rccl& R = Rccl::get();
            if (!R.ok()) throw std::runtime_error("MultiGPU: librccl.so.1 could not be loaded (needed for more than one GPU)");
            if (impl->comms.empty()) {
                std::vector<int> devs;
                for (size_t g = 0; g < nGpus; g++) devs.push_back(devices[impl->leader[g]]);
                impl->comms.assign(nGpus, nullptr);
                const int rc = R.CommInitAll(impl->comms.data(), (int)nGpus, devs.data());
                if (rc != 0) { impl->comms.clear(); throw std::runtime_error(std::string("ncclCommInitAll: ") + (R.GetErrorString ? R.GetErrorString(rc) : "failed")); }
                impl->streams.assign(nGpus, nullptr);
                for (size_t g = 0; g < nGpus; g++) {
                    if (hipSetDevice(devs[g]) != hipSuccess || hipStreamCreate(&impl->streams[g]) != hipSuccess) throw std::runtime_error("MultiGPU: stream creation failed");
                }
            }
            for (uint32_t img = 0; img < nImages; img++) {
                std::vector<std::string> rerr(nGpus);
                auto red = [&](size_t g) {
                    const size_t i = impl->leader[g];
                    void* p = nullptr;
                    if (rt_image_device_ptr(shaders[i]->ID, img, &p) != RT_OK) { rerr[g] = rt_last_error(shaders[i]->ID); return; }
                    if (hipSetDevice(devices[i]) != hipSuccess) { rerr[g] = "hipSetDevice failed"; return; }
                    const int rc = R.Reduce(p, p, (size_t)width * height * 4, Rccl::kFloat, Rccl::kSum, 0, impl->comms[g], impl->streams[g]);
                    if (rc != 0) { rerr[g] = std::string("ncclReduce: ") + (R.GetErrorString ? R.GetErrorString(rc) : "failed"); return; }
                    // the next frame clears this image on the context's own stream: wait for the reduce here
                    if (hipStreamSynchronize(impl->streams[g]) != hipSuccess) rerr[g] = "hipStreamSynchronize after ncclReduce failed";
                };
                std::vector<std::thread> th;
                for (size_t g = 0; g < nGpus; g++) th.emplace_back(red, g);
                for (auto& t : th) t.join();
                for (size_t g = 0; g < nGpus; g++) if (!rerr[g].empty()) throw std::runtime_error("GPU " + std::to_string(devices[impl->leader[g]]) + ": " + rerr[g]);
            }
            lastUsedRccl = true;
        }
    }
    // Every context runs what context 0's per-scene kernel trial kept (the trial is host-clocked, so contexts could settle
    // differently: same images, different shard times).  As distributed.agree_on_kernels does across processes.
    framesRendered++;
    if (world > 1 && !kernelsAgreed) {
        rt_stats st;
        if (rt_get_stats(shaders[0]->ID, &st) == RT_OK && st.kernel_choice_settled && st.kernel_choice >= 0) {
            for (size_t i = 1; i < shaders.size(); i++) shaders[i]->setInt("kernelChoice", st.kernel_choice);
            kernelsAgreed = true;
        }
    }
}

}  // namespace rt_host

using namespace rt_host;

namespace {
template <typename F>
rt_status guarded(F&& f) {
    try { f(); return RT_OK; }
    catch (const std::exception& e) { set_last_error(e.what()); return RT_ERR_SCENE; }
    catch (...) { set_last_error("unknown error"); return RT_ERR_SCENE; }
}
}  // namespace

extern "C" {

// ---- multi-GPU driver --------------------------------------------------------------------------------------------------
struct rth_multi { MultiGPU* m = nullptr; };

rt_status rth_multi_create(rth_multi** out, const int32_t* devices, int32_t n) {
    if (!out || !devices || n <= 0) { set_last_error("bad argument"); return RT_ERR_INVALID; }
    *out = nullptr;
    return guarded([&] {
        std::vector<int> d(devices, devices + n);
        rth_multi* h = new rth_multi();
        try { h->m = new MultiGPU(d); } catch (...) { delete h; throw; }
        *out = h;
    });
}
void rth_multi_destroy(rth_multi* h) { if (h) { delete h->m; delete h; } }
int32_t rth_multi_size(const rth_multi* h) { return h && h->m ? (int32_t)h->m->size() : 0; }
rt_ctx* rth_multi_ctx(rth_multi* h, int32_t i) { return (h && h->m && i >= 0 && (size_t)i < h->m->size()) ? h->m->shader((size_t)i)->ID : nullptr; }
rt_status rth_multi_load_scene(rth_multi* h, const char* models_root, int32_t scene_index) {
    if (!h || !h->m || !models_root) { set_last_error("NULL argument"); return RT_ERR_INVALID; }
    rt_status s = rth_load_scene(h->m->shader(0)->ID, models_root, scene_index);
    if (s != RT_OK) return s;
    return guarded([&] { h->m->ReplicateScene(); });
}
rt_status rth_multi_load_models(rth_multi* h, const char* const* paths, int32_t n_models, const float* lights, int32_t n_lights) {
    if (!h || !h->m) { set_last_error("NULL argument"); return RT_ERR_INVALID; }
    rt_status s = rth_load_models(h->m->shader(0)->ID, paths, n_models, lights, n_lights);
    if (s != RT_OK) return s;
    return guarded([&] { h->m->ReplicateScene(); });
}
rt_status rth_multi_set_int(rth_multi* h, const char* name, int32_t v) {
    if (!h || !h->m || !name) { set_last_error("NULL argument"); return RT_ERR_INVALID; }
    return guarded([&] { h->m->setInt(name, v); });
}
rt_status rth_multi_resize(rth_multi* h, uint32_t w, uint32_t hh) {
    if (!h || !h->m) { set_last_error("NULL argument"); return RT_ERR_INVALID; }
    return guarded([&] { h->m->resize(w, hh); });
}
rt_status rth_multi_set_camera(rth_multi* h, const float position[3], const float orientation[3], float w, float hh, float fov) {
    if (!h || !h->m || !position || !orientation) { set_last_error("NULL argument"); return RT_ERR_INVALID; }
    return guarded([&] {
        Camera cam((int)w, (int)hh, vec3(position[0], position[1], position[2]));
        cam.Orientation = vec3(orientation[0], orientation[1], orientation[2]);
        cam.FOVdeg = fov;
        vec2 ss; ss.x = w; ss.y = hh;
        h->m->SetFrameUniforms(&cam, ss);
    });
}
rt_status rth_multi_render_frame(rth_multi* h, uint32_t spp, int32_t band_ranks, int32_t merge_aovs) {
    if (!h || !h->m) { set_last_error("NULL argument"); return RT_ERR_INVALID; }
    return guarded([&] { h->m->RenderFrame(spp, band_ranks < 1 ? 1 : band_ranks, merge_aovs != 0); });
}
int32_t rth_multi_used_rccl(const rth_multi* h) { return h && h->m && h->m->usedRccl() ? 1 : 0; }
rt_status rth_partition(uint32_t width, uint32_t height, uint32_t spp, int32_t rank, int32_t world, int32_t band_ranks, uint32_t out[6]) {
    if (!out) { set_last_error("NULL argument"); return RT_ERR_INVALID; }
    rt_status s = guarded([&] {
        const Shard sh = partition(width, height, spp, rank, world, band_ranks);
        out[0] = sh.x; out[1] = sh.y; out[2] = sh.w; out[3] = sh.h; out[4] = sh.r0; out[5] = sh.r1;
    });
    return s == RT_OK ? RT_OK : RT_ERR_INVALID;
}

}  // extern "C"
