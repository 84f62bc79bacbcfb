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

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
