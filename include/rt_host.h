/*
 * rt_host.h — C ABI of the host-side callers that sit above the integrator boundary (also in
 * librt_core.so).  These are the reference's own C++ callers rebuilt GL-free so the whole
 * load -> BVH -> upload -> render -> read-back path runs headless; each entry point cites the
 * reference code it stands for (paths relative to /root/reference/Raytracing/).
 * The C++ classes behind them (raytracing_amd/csrc/host/rt_host.hpp) keep the reference's names:
 * Model, Light, Scene, Camera, ComputeShader, SSBO, Renderer.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stdint.h>
#include "rt_core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Message of the last failing rth_* call on this thread. */
const char* rth_last_error(void);

/* Scene::LoadScene(computeShader, sceneIndex) (Scene.cpp:15-62) + Scene::LoadGPUData (Scene.cpp:64-76):
 * unloads the previous scene, creates the lights and Models of scene `scene_index` (1..7) with model paths
 * resolved under `models_root` (the reference's "models/" directory), flattens them (Model::BVH,
 * model.cpp:532-582), builds the SAH BVH (model.cpp:417-530), uploads slots 0-5 through rt_upload_buffer and
 * sets numLights / hasSun.  Textures are decoded (PNG) and registered with rt_create_texture.
 * ctx == NULL is a dry run for hosts without a GPU: nothing is uploaded, the six buffers are only kept for
 * rth_scene_buffer, and texture handles are numbered 1, 2, ... in creation order. */
rt_status rth_load_scene(rt_ctx* ctx, const char* models_root, int32_t scene_index);

/* Same pipeline for an explicit model list + lights (what Scene.cpp's switch hard-codes):
 * lights = n_lights records of 8 floats {loc.xyz, dir.xyz, type (0 point / 1 sun), strength}. */
rt_status rth_load_models(rt_ctx* ctx, const char* const* gltf_paths, int32_t n_models,
                          const float* lights, int32_t n_lights);

/* Host copies of the six buffers of the last rth_load_* call on this thread (reference layouts), for
 * inspection and parity tests.  Pointers stay valid until the next rth_load_* call. */
rt_status rth_scene_buffer(uint32_t slot, const void** data, uint64_t* bytes);
/* Seconds spent in the last load: [0] GLTF parse+flatten, [1] BVH build, [2] upload. */
rt_status rth_load_timings(double out[3]);

/* Model::buildBVH + convertToGPU alone (model.cpp:467-530): `n` 128-byte Triangle records ->
 * malloc'ed array of 48-byte GPUBoundingBox records (free with rth_free). */
rt_status rth_build_bvh(const void* triangles, uint64_t n, void** nodes_out, uint64_t* n_nodes_out);
void rth_free(void* p);

/* The per-frame camera uniforms of main.cpp:109-123: camPos, camForward = normalize(Orientation),
 * camRight = normalize(cross(forward, (0,1,0))), camUp = cross(right, forward), aspect = w / h, fov. */
rt_status rth_set_camera(rt_ctx* ctx, const float position[3], const float orientation[3],
                         float screen_w, float screen_h, float fov_deg);
/* Renderer::UpdateCamera (Renderer.cpp:187-193) for frame index `frame`. */
void rth_orbit_camera(int32_t frame, float position_out[3], float orientation_out[3]);

/* Renderer::NextFrame's pixel conversion (Renderer.cpp:95-124): reads the colour image, clamps to [0,1],
 * truncates to 8 bits, flips vertically; rgba8_out = W*H*4 bytes, top row first. */
rt_status rth_frame_rgba8(rt_ctx* ctx, uint32_t width, uint32_t height, uint8_t* rgba8_out);

/* Texture file (PNG or JPEG, told apart by content as stbi_load does) -> RGBA8 as Texture.cpp:5-34 uploads it (row 0 =
 * first row of the file; 3 channels get A = 255, 1 channel lands in R alone).  malloc'ed; rth_free.
 * rth_decode_png is the older name of the same call. */
rt_status rth_decode_image(const char* path, uint8_t** rgba8_out, uint32_t* w, uint32_t* h);
rt_status rth_decode_png(const char* path, uint8_t** rgba8_out, uint32_t* w, uint32_t* h);

/* ---- Multi-GPU frame driver (no reference counterpart: the reference is single-GPU, one GL context) --------------------
 * The frame loop of main.cpp:106-141 / Renderer.cpp:91-193 on N GPUs of one node from ONE process: one rt_ctx per device
 * with the scene replicated, one host thread per device that renders its (row band x sample range) shard with
 * rt_render_region, then ONE collective: ncclReduce(sum, float32) of the colour image onto the first device (RCCL over
 * xGMI; librccl.so.1 is loaded on first use, and a group of one device never touches it).  Devices may repeat in the
 * list (several contexts on one GPU: a rehearsal on a single-GPU box); images that share a GPU are merged with
 * rt_add_image and RCCL runs among the distinct GPUs only.  The Python driver of the same partition is
 * raytracing_amd/distributed.py (one process per GPU, torch.distributed). */
typedef struct rth_multi rth_multi;
rt_status rth_multi_create(rth_multi** out, const int32_t* devices, int32_t n_devices);
void rth_multi_destroy(rth_multi* m);
int32_t rth_multi_size(const rth_multi* m);
rt_ctx* rth_multi_ctx(rth_multi* m, int32_t i);          /* context i (0 = where the merged frame lands) */
/* Scene::LoadScene on context 0, then the same six buffers, textures (same handles) and numLights / hasSun on the others. */
rt_status rth_multi_load_scene(rth_multi* m, const char* models_root, int32_t scene_index);
rt_status rth_multi_load_models(rth_multi* m, const char* const* gltf_paths, int32_t n_models, const float* lights, int32_t n_lights);
/* rt_set_int / rt_resize / rth_set_camera on every context. */
rt_status rth_multi_set_int(rth_multi* m, const char* name, int32_t v);
rt_status rth_multi_resize(rth_multi* m, uint32_t w, uint32_t h);
rt_status rth_multi_set_camera(rth_multi* m, const float position[3], const float orientation[3], float screen_w, float screen_h, float fov_deg);
/* One frame of samplesPerPixel = spp on all contexts: `band_ranks` row bands x (N / band_ranks) sample ranges
 * (rth_partition), merged onto context 0 (colour; the three AOV images too when merge_aovs != 0).  After a few frames
 * every context is pinned to context 0's settled kernel choice (rt_stats.kernel_choice), so that shards cost the same. */
rt_status rth_multi_render_frame(rth_multi* m, uint32_t spp, int32_t band_ranks, int32_t merge_aovs);
/* 1 when the last rth_multi_render_frame reduced through RCCL, 0 when no collective was needed (one distinct GPU). */
int32_t rth_multi_used_rccl(const rth_multi* m);
/* The partition itself (pure index math, the same as raytracing_amd.distributed.partition): shard of `rank` of `world`
 * -> out = {x, y, w, h, r0, r1} for rt_render_region; w * h == 0 or r0 == r1 means an empty shard. */
rt_status rth_partition(uint32_t width, uint32_t height, uint32_t spp, int32_t rank, int32_t world, int32_t band_ranks, uint32_t out[6]);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
