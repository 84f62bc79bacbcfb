/*
 * rt_core.h — C ABI of librt_core.so, the MI355X (gfx950) path-tracing core.
 *
 * Drop-in boundary: the reference (bspafford/Raytracing) has no plugin/FFI interface; its seam to
 * the integrator is the set of OpenGL calls in SSBO.cpp / computeShader.cpp / main.cpp.  Every entry
 * point below replaces one of those calls one-for-one (reference file:line cited per function;
 * paths are relative to /root/reference/Raytracing/).  INTEGRATION.md shows the edits a maintainer
 * makes in SSBO.cpp, computeShader.cpp, Texture.cpp and main.cpp to bind them.
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  Every function returns rt_status
 * (0 = RT_OK); rt_last_error() gives the message.  The callee copies every input before returning
 * (the reference frees its host arrays right after SSBO::Bind, model.cpp:607-610).  A context is
 * bound to one GPU and is not thread-safe (the reference drives GL from one thread).  There is no
 * CPU fallback: without a HIP device rt_create fails with RT_ERR_NO_DEVICE.
 */
#ifndef RT_CORE_H
#define RT_CORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t rt_status;
enum {
    RT_OK = 0,
    RT_ERR_NO_DEVICE = 1,      /* no HIP device / hipSetDevice failed */
    RT_ERR_INVALID = 2,        /* bad argument (NULL, slot/image out of range, size not a record multiple) */
    RT_ERR_NOT_READY = 3,      /* render requested before slots 0..5 and the image size were provided */
    RT_ERR_HIP = 4,            /* a HIP call or kernel failed; message has hipGetErrorString */
    RT_ERR_SCENE = 5           /* inconsistent scene buffers (index out of range, cyclic BVH ...) */
};

typedef struct rt_ctx rt_ctx;

typedef struct rt_config {
    int32_t device;            /* HIP device ordinal */
    int32_t rank, world_size;  /* informational (multi-GPU partition is driven by rt_render_region) */
    int32_t reserved;
} rt_config;

/* Buffer slots = the SSBO binding points of shader.comp:78-83. */
enum {
    RT_SLOT_VERTICES = 0,      /* Vertex[]          80 B  VBO.h:9-17                         */
    RT_SLOT_MESH_MATRICES = 1, /* mat4[]            64 B  column-major, model.cpp:562-563    */
    RT_SLOT_MATERIALS = 2,     /* MaterialData[]    80 B  model.h:8-39                       */
    RT_SLOT_BVH_NODES = 3,     /* GPUBoundingBox[]  48 B  model.h:130-142, DFS pre-order     */
    RT_SLOT_TRIANGLES = 4,     /* Triangle[]       128 B  model.h:41-70, original order      */
    RT_SLOT_LIGHTS = 5         /* GPULight[]        48 B  light.h:14-23                      */
};

/* Output images = image units 0..3 of shader.comp:9-12 (RGBA32F, row 0 = bottom). */
enum { RT_IMAGE_COLOR = 0, RT_IMAGE_ALBEDO = 1, RT_IMAGE_NORMAL = 2, RT_IMAGE_DEPTH = 3 };

/* Replaces ComputeShader::ComputeShader(const char*) (computeShader.cpp:7-23): creates the
 * integrator "program" on one GPU.  cfg may be NULL (device 0). */
rt_status rt_create(rt_ctx** out, const rt_config* cfg);
/* Replaces ComputeShader::Delete (computeShader.cpp:29-31).  Frees all device memory. */
void rt_destroy(rt_ctx* ctx);
/* Message of the last failing call on ctx (ctx may be NULL for rt_create failures). */
const char* rt_last_error(const rt_ctx* ctx);

/* Replaces SSBO::Bind(const void* data, GLuint size, GLuint slot) (SSBO.cpp:5-17): copies `bytes`
 * bytes now; re-binding a slot replaces its content.  The reference record layouts above are the
 * input format; the compacted device layout is derived lazily at the next render. */
rt_status rt_upload_buffer(rt_ctx* ctx, uint32_t slot, const void* data, uint64_t bytes);

/* Replace ComputeShader::setInt/setFloat/setVec2/setVec3/setVec4/setMat4 (computeShader.cpp:53-75):
 * by-name, sticky.  Live names (shader.comp:85-102): numLights, hasSun, camPos, camForward, camRight,
 * camUp, aspect, fov, samplesPerPixel, maxBounces, offset.  Unknown names (projection, invProj, view,
 * invView, time ...) are accepted and ignored, as glGetUniformLocation == -1 is in GL.
 * Extension names (no reference counterpart): traversalMode (0 = culled/ordered, 1 = reference-order
 * exhaustive with the shader's 64-entry stack; 0 becomes 1 by itself for a scene on which culling would not give the
 * reference's result: boxes that do not contain their triangles, trees 63 or more levels deep), timeKernels (1 = HIP-event timing per kernel class into rt_stats), countVisits (1 = run the
 * instrumented traversal kernels that count node visits / triangle tests), samplesPerPass (samples of a pixel
 * traced concurrently; results do not depend on it), sampleGroup (0..6; default -1 = 4, or 6 where the BVH is far larger than the L2s: a wave of the camera-ray launch holds
 * 2^sampleGroup samples of 64 >> sampleGroup neighbouring pixels; results do not depend on it), logDenseBounces (0 automatic: 4 when maxBounces > 8, else all) and lateLogPercent (default 25): the
 * contribution log keeps an entry per slot for that many bounces and, for the later ones, per queue position, as many as that
 * share of the slots (a frame that needs more is rendered again with a dense log; results do not depend on either),
 * pathMemPercent (default 75: share of device memory the path state may take; caps the
 * samples per pass), blocksPerCU (persistent grid size), passStreams (1..8, default 4: consecutive
 * passes of samplesPerPass samples run on that many HIP streams with their own path buffers and overlap; the per-pixel
 * sum is still replayed in sample order, so results do not depend on it; unless the caller sets passStreams or
 * samplesPerPass, the frames after the kernel trial also measure half the streams with twice the samples per pass and
 * keep that pass shape for the scene if it is 2 % faster), refill (-1 automatic / 0 / 1: persistent
 * traversal waves that hand finished lanes new rays; automatic = the first four sizeable frames after a scene upload
 * are rendered with the four combinations (refill or not, for closest-hit and for shadow rays) and the fastest is kept;
 * until then closest-hit rays of bounces >= 1 use them, and shadow rays do where the BVH is far larger than the L2s), kernelChoice (-1 / 0..7: bits 0 and 1 that combination, bit 2 the pass
 * shape (1: half the streams, twice the samples per pass), whatever
 * was measured, reported back in rt_stats.kernel_choice; a multi-GPU job gives every rank rank 0's), fusePairs (1 / 0: the
 * shadow rays of a bounce and the closest-hit rays of the next are traced by one launch), wideNodes (-1 / 0 / 1: the refill
 * kernels walk two-level nodes derived from the same tree, by default for BVHs far larger than the L2s; same images),
 * packetBounces (-1 / n: the sun's shadow rays of the first n bounces are traced 64 to a wave through scalar loads; -1 = bounce
 * 0 where the compact BVH is at most 1 MB), refillFromBounce, autoPass, sunKernel, tuneCache, refillIdle, refillNodeMin,
 * refillTriMin, refillNodeLow, refillDefer, refillBlocksPerCU (all in INTEGRATION.md section 7, with defaults). None of them
 * changes an image. */
rt_status rt_set_int(rt_ctx* ctx, const char* name, int32_t v);
rt_status rt_set_float(rt_ctx* ctx, const char* name, float v);
rt_status rt_set_vec2(rt_ctx* ctx, const char* name, const float v[2]);
rt_status rt_set_vec3(rt_ctx* ctx, const char* name, const float v[3]);
rt_status rt_set_vec4(rt_ctx* ctx, const char* name, const float v[4]);
rt_status rt_set_mat4(rt_ctx* ctx, const char* name, const float m[16]);

/* Replaces Texture::Texture(GLuint slot, glm::vec2 size) x4 (Texture.cpp:36-47, main.cpp:85-88) and the
 * resize path main.cpp:293-306: (re)allocates the four W x H RGBA32F output images, zero-filled. */
rt_status rt_resize(rt_ctx* ctx, uint32_t width, uint32_t height);

/* Replaces Texture::Texture(image, type, slot) + glGetTextureHandleARB/glMakeTextureHandleResidentARB
 * (Texture.cpp:5-34, model.cpp:546-559): uploads a decoded RGBA8 image (row 0 first, as stbi returns it)
 * and returns the 64-bit handle to store in MaterialData.  Sampling = LINEAR, REPEAT, level 0. */
rt_status rt_create_texture(rt_ctx* ctx, const uint8_t* rgba8, uint32_t width, uint32_t height, uint64_t* handle_out);

/* Replaces the 64x64-tile glDispatchCompute loop + glMemoryBarrier (main.cpp:126-141): traces the whole
 * frame (all samplesPerPixel, all bounces) into the four images and returns when they are complete. */
rt_status rt_render_frame(rt_ctx* ctx);
/* Partition form used by the multi-GPU driver (no reference counterpart; the reference is single-GPU):
 * traces pixels [x,x+w) x [y,y+h) and samples [spp_begin, spp_end) of samplesPerPixel.  Pixels outside the
 * region are zero in all four images; the AOV images (albedo, normal, depth) are written when spp_begin == 0 and
 * are all-zero otherwise, and the colour image's alpha is 1 when spp_begin == 0 and 0 otherwise, so partial renders
 * merge with one element-wise sum into exactly the single-call frame's layout (alpha 1 in all four images). */
rt_status rt_render_region(rt_ctx* ctx, uint32_t x, uint32_t y, uint32_t w, uint32_t h,
                           uint32_t spp_begin, uint32_t spp_end);

/* Replaces glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_FLOAT, dst) (main.cpp:157,163,169; Renderer.cpp:100):
 * blocking copy of W*H*4 floats, row 0 = bottom. */
rt_status rt_read_image(rt_ctx* ctx, uint32_t image, float* dst_rgba);
/* Replaces glTexSubImage2D write-back after denoise (main.cpp:184). */
rt_status rt_write_image(rt_ctx* ctx, uint32_t image, const float* src_rgba);
/* The denoiser hand-off of main.cpp:152-184 without its host loops: the reference reads each RGBA image back, strips the
 * alpha on the CPU (rgbaTOrgb, main.cpp:158,164,170) into OIDN's Float3 buffers, and re-expands the denoised colour to
 * RGBA with alpha 1 before glTexSubImage2D (main.cpp:176-184).  These two do the repacking on the GPU and move W*H*3
 * floats: rt_read_image_rgb fills an OIDN Float3 buffer, rt_write_image_rgb takes one back (alpha := 1). */
rt_status rt_read_image_rgb(rt_ctx* ctx, uint32_t image, float* dst_rgb);
rt_status rt_write_image_rgb(rt_ctx* ctx, uint32_t image, const float* src_rgb);
/* Replaces the read-back + 8-bit conversion of Renderer::NextFrame (Renderer.cpp:95-107) and the vertical flip of its
 * frame copy (121-127): the colour image clamped to [0, 1], * 255, truncated to unsigned char, W*H*4 bytes, TOP row
 * first.  Converted on the GPU; a quarter of the bytes of rt_read_image cross PCIe. */
rt_status rt_read_frame_rgba8(rt_ctx* ctx, uint8_t* dst_rgba8, uint64_t dst_bytes /* must be W*H*4 */);
/* Device address of an image (W*H*4 floats) for zero-copy consumers on the same GPU (RCCL reduce,
 * device-side denoisers).  No reference counterpart. */
rt_status rt_image_device_ptr(rt_ctx* ctx, uint32_t image, void** dev_ptr);

/* dst image += an image of the same size that lives on the SAME GPU (another context's rt_image_device_ptr): the merge step
 * of partial renders (row bands / sample ranges, rt_render_region) when two contexts share one GPU.  Across GPUs the merge
 * is the RCCL reduce (rt_host.h, rth_multi_render_frame).  No reference counterpart. */
rt_status rt_add_image(rt_ctx* ctx, uint32_t image, const void* src_dev_ptr);

/* Counters of the last rt_render_* call (no reference counterpart; feeds the Mrays/s metric). */
typedef struct rt_stats {
    uint64_t rays_primary;     /* lineCast calls at bounce 0                                   */
    uint64_t rays_bounce;      /* lineCast calls at bounce > 0                                 */
    uint64_t rays_shadow;      /* shadow lineCast calls actually traced                        */
    uint64_t rays_shadow_skipped; /* shadow casts the reference issues whose result cannot matter (N.L <= 0) */
    uint64_t node_visits;      /* closest-hit kernel: 64-byte node records fetched   (countVisits = 1) */
    uint64_t tri_tests;        /* closest-hit kernel: 48-byte triangle records tested (countVisits = 1) */
    uint64_t shadow_node_visits; /* same two counters for the shadow kernel */
    uint64_t shadow_tri_tests;
    uint64_t max_ray_steps;    /* most node visits + triangle tests by a single closest-hit ray (countVisits = 1) */
    double   ms_total;         /* wall time of the render call, host clock                     */
    double   ms_extend;        /* summed HIP-event time of the closest-hit kernels (timeKernels = 1) */
    double   ms_shade;         /* ... of the shading kernels                                   */
    double   ms_shadow;        /* ... of the shadow (any-hit) kernels                          */
    uint64_t launches_extend;  /* kernel launches behind ms_extend                             */
    uint64_t launches_shade;
    uint64_t launches_shadow;
    uint64_t bounce_node_visits; /* node_visits / tri_tests of bounces >= 1 only (countVisits = 1)            */
    uint64_t bounce_tri_tests;
    uint64_t wave_node_steps;  /* wave-level executions of the node step / triangle step by those kernels:   */
    uint64_t wave_tri_steps;   /* lane utilisation of a step = visits / (64 x executions)                     */
    double   ms_extend_primary;      /* the part of ms_extend / launches_extend spent on camera rays (bounce 0)       */
    uint64_t launches_extend_primary;
    int32_t  kernel_choice;    /* traversal kernels this call ran: bit 0 = refill kernel for closest-hit rays of bounces >= 1,
                                  bit 1 = for shadow rays, bit 2 = the pass shape the next call will use (half the streams, twice the
                                  samples per pass); -2 = reference-order walk (traversalMode 1 or forced) */
    int32_t  kernel_choice_settled;  /* 1: fixed by a knob or by the finished per-scene trials (kernels, then pass shape); 0: a trial frame or the default */
    uint64_t launches_pair;    /* launches that traced the shadow rays of a bounce and the closest-hit rays of the next together
                                  (k_trace_pair, knob fusePairs); they are counted and timed under launches_extend / ms_extend */
    int32_t  wide_nodes;       /* 1: the refill kernels of this call walked two-level nodes (knob wideNodes and a tree that qualifies) */
    int32_t  wide_depth;       /* two-level steps on the longest root-to-leaf path (0: not derived) */
    int32_t  log_dense_bounces;  /* bounces whose contribution log holds an entry per slot; later bounces log by queue position
                                    (knobs logDenseBounces, lateLogPercent; == maxBounces: the whole log is dense) */
    int32_t  log_fallbacks;      /* frames of this context rendered twice because the compact part of the log overflowed */
} rt_stats;
rt_status rt_get_stats(rt_ctx* ctx, rt_stats* out);

/* Debug / parity taps (no reference counterpart): primary-hit triangle ids (reference numbering, -1 = miss)
 * and hit locations of sample 0, W*H int32 / W*H*3 floats, valid after a render whose sample range contains 0. */
rt_status rt_read_primary_hits(rt_ctx* ctx, int32_t* tri_ids, float* hit_locs);
/* Closest-hit query on caller rays with the same kernel the integrator uses (parity tests):
 * origins/dirs n*3 floats (host) -> tri_ids n int32 (reference numbering), hit_locs n*3 floats. */
rt_status rt_trace_rays(rt_ctx* ctx, const float* origins, const float* dirs, uint64_t n,
                        int32_t* tri_ids, float* hit_locs);

/* rt_trace_rays for callers that cannot allocate: tight serving loops, harnesses that forbid hipMalloc, code that must
 * not disturb the device heap.  No counterpart in the reference.  (Not capturable into a HIP graph as it stands: it copies
 * from / to pageable host memory and synchronises the context's stream before it returns, also on its error paths.)  Same arguments, same kernels and bit-identical results, but the
 * ray queue, hit records, queue counter, traversal-stack overflow and the staged inputs / outputs live in a file-scope
 * device array of the library (about 8.4 MB, held by every GPU the library runs on until the process ends) instead of
 * hipMalloc'ed buffers, and the call makes no hipMalloc / hipFree / allocating library call.  Limits: n <=
 * RT_NOALLOC_MAX_RAYS per call (more: RT_ERR_INVALID); the scene must already be resident, i.e. an rt_render_* or
 * rt_trace_rays call has run since the last upload of slots 0..4 (otherwise RT_ERR_NOT_READY: deriving the compact scene
 * allocates); hit_locs is required.  All contexts on one GPU share that GPU's array and it keeps the last call's
 * contents (the call clears what must start at zero): calls that overlap on one GPU race, so serialise them per GPU. */
#define RT_NOALLOC_MAX_RAYS 16384
rt_status rt_trace_rays_noalloc(rt_ctx* ctx, const float* origins, const float* dirs, uint64_t n,
                                int32_t* tri_ids, float* hit_locs);

#ifdef __cplusplus
}
#endif
#endif /* RT_CORE_H */
