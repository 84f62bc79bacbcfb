// gfx950 wavefront path-tracing kernels (included by rt_core.hip).
//
// Pipeline per pass of S samples (S paths per pixel in flight; DESIGN.md "Wavefront pipeline"):
//   (primary_ray)       shader.comp:309-324   camera rays of samples [rBase, rBase+S), regenerated from the slot index
//   for bounce b:
//     k_extend          shader.comp:243-290   closest hit (lineCast) over queue b
//     k_shade           shader.comp:329-452,524-531  miss/ambient, surface fetch, dielectric/GGX bounce,
//                                              light evaluation -> contribution log, queue b+1
//     k_shadow          shader.comp:453-484   shadow lineCast per pending light -> confirms / cancels the log entry
//   k_resolve           shader.comp:332,521   replays the log into the pixel colour in the reference's order
//                                              (sample, bounce, light), so float sums are bit-identical to the serial shader
//   k_finalize          shader.comp:533-536   image stores
// Compaction between stages: wave64 ballot + mbcnt prefix, wave totals through LDS, one atomic per queue per 1024 paths.
// Consecutive passes run on several streams with their own PathBuffers (rt_core.hip); only k_resolve is ordered.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include "rt_types.h"
#include "rt_math.h"

namespace rt {

#define RT_BLOCK 256

// An entry of the contribution log (PathBuffers::contrib): three floats, 12 bytes.
#ifdef RT_EXP_LOG16
typedef float4 LogEntry;
__device__ __forceinline__ LogEntry log_entry(float x, float y, float z) { return make_float4(x, y, z, 0.f); }
#else
typedef float3 LogEntry;
__device__ __forceinline__ LogEntry log_entry(float x, float y, float z) { return make_float3(x, y, z); }
#endif

struct PathBuffers {
    // Per-path state lives at the path's POSITION IN ITS BOUNCE'S QUEUE, not at its slot: a bounce's kernels then read
    // and write it with dense, coalesced accesses instead of gathers through a queue of slot ids (the scattered
    // 16-byte accesses into multi-GB arrays were HBM round trips on every ray).  Two buffers alternate by bounce parity.
    float4* ray[2];      // entry i: [2i] = (o.xyz, d.x), [2i+1] = (d.y, d.z, flag bits (bit0 refracting), slot bits)
    float4* rayS[2];     // entry i: strength.xyz, slot bits (a path that left the scene needs nothing else of its state)
    float2* hit;         // entry i of the bounce being traced: (t, reference triangle index bits: -1 miss, -2 dead slot);
                         // hitLoc = o + d * t is recomputed by k_shade
    float4* shadowRec;   // shadow-queue entry j: (origin.xyz, slot bits)
    float4* shadeLoc;    // [slot] hitLoc of the bounce being shaded, written only when a point light needs it (k_shadow)
    // Indexed by slot (one per sample of the pass and pixel: slot_split), because k_resolve replays them per pixel:
    // Contribution log [maxBounces][lightSlots][nslots], 12 bytes per entry: what `color +=` adds at that bounce for that
    // light, or +0 where nothing is added (N.L <= 0, a contribution of exactly zero, an occluded light: k_shade writes the
    // contribution, the shadow kernels overwrite it with zeros when the light is occluded).  The replay adds every entry of a
    // path's bounces: x + (+0) == x bit for bit for every x the sum can hold (it starts at +0 and a sum is -0 only if both
    // operands are, so it never is -0; NaN stays NaN), so an entry needs no state word.
    LogEntry* contrib;
    uint32_t* depth;     // [nslots] bounces shaded by the slot's path (log entries [0, depth) are valid)
    // Deep-bounce renders keep only the first `denseBounces` planes of the log per slot; later bounces, which few paths
    // reach, log at their path's POSITION IN THAT BOUNCE'S QUEUE (lateCap entries per bounce) and a slot finds its entries
    // through lateHead (its position at bounce denseBounces) and lateNext (from a position to the next bounce's).
    LogEntry* late;        // [maxBounces - denseBounces][lightSlots][lateCap]
    uint32_t* lateNext;  // [maxBounces - denseBounces][lateCap]
    uint32_t* lateHead;  // [nslots]
    uint32_t denseBounces, lateCap;      // denseBounces >= maxBounces: the whole log is dense
    float4* accum;       // [slotsPerSample] running pixel colour (the shader's `color`), survives passes, shared by all sets
    uint32_t* counters;  // [0..maxB] path-queue sizes per bounce, [maxB+1 .. 2maxB+1] shadow-queue sizes
    uint2* overflow;     // traversal-stack entries beyond the LDS part: entry k of thread g at overflow[k*overflowStride + g]
    uint32_t overflowStride, overflowDepth;
    uint32_t nslots;     // slots in flight = samples per pass x slotsPerSample
    uint32_t slotsPerSample, lightSlots;   // lightSlots = max(1, numLights)
    uint32_t grpShift;   // a group of 64 slots holds 2^grpShift samples of 64 >> grpShift pixels (slot_split)
};

__device__ __forceinline__ void unpack_ray(float4 a, float4 b, v3& o, v3& d, uint32_t& flags, uint32_t& slot) {
    o = mk(a.x, a.y, a.z); d = mk(a.w, b.x, b.y); flags = __float_as_uint(b.z); slot = __float_as_uint(b.w);
}

// The log entries of one bounce: entry of light l and log index x at base[l * stride + x], where x is the path's slot
// for a bounce of the dense part and its position in the bounce's queue for a later one.
struct LogPlane { LogEntry* base; size_t stride; };
__device__ __forceinline__ LogPlane log_plane(const PathBuffers& P, int bounce) {
    LogPlane L;
    if ((uint32_t)bounce < P.denseBounces) { L.base = P.contrib + (size_t)bounce * P.lightSlots * P.nslots; L.stride = P.nslots; }
    else { L.base = P.late + (size_t)((uint32_t)bounce - P.denseBounces) * P.lightSlots * P.lateCap; L.stride = P.lateCap; }
    return L;
}

// ---------------------------------------------------------------------------------------------
// Block-aggregated stream compaction for RT_ITEMS flags per thread and NQ output queues:
// wave64 ballot + mbcnt prefix inside each wave, wave totals combined through LDS, ONE atomic per queue per
// block (a single counter word sustains only ~88 atomics/us on MI355X, so per-wave atomics would bound the
// kernel).  Returns the output index of item k of this thread in idx[q][k] (valid where flag set).
#ifndef RT_ITEMS
#define RT_ITEMS 4
#endif
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 1
#endif
template <int NQ>
__device__ __forceinline__ void block_append(const bool (&flag)[NQ][RT_ITEMS], uint32_t* const (&counter)[NQ],
                                             uint32_t (&idx)[NQ][RT_ITEMS], uint32_t* ldsScratch /* >= NQ*8 words */) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t waveTotal[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < RT_ITEMS; k++) {
            unsigned long long m = __ballot(flag[q][k]);
            uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            idx[q][k] = run + pre;
            run += (uint32_t)__popcll(m);
        }
        waveTotal[q] = run;
        if (lane == 0) ldsScratch[q * 8 + wave] = run;
    }
    __syncthreads();
    if (threadIdx.x < NQ) {
        const int q = threadIdx.x;
        uint32_t t = 0;
        for (uint32_t w = 0; w < RT_BLOCK / 64; w++) { uint32_t c = ldsScratch[q * 8 + w]; ldsScratch[q * 8 + w] = t; t += c; }
        uint32_t base = t ? atomicAdd(counter[q], t) : 0u;
        ldsScratch[q * 8 + 4] = base;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        uint32_t off = ldsScratch[q * 8 + 4] + ldsScratch[q * 8 + wave];
#pragma unroll
        for (int k = 0; k < RT_ITEMS; k++) idx[q][k] += off;
    }
    __syncthreads();   // ldsScratch is reused by the next chunk
    (void)waveTotal;
}

// Light l of slot 5: the first RT_MAX_LIGHTS travel in the kernel argument segment (uniform index: scalar loads),
// further ones are read from the uploaded buffer.
// (The buffer read is volatile so that it stays behind its branch: written as a select, the compiler fetched the
// global record per lane for every light of every path, which cost k_shade 12 %.)
__device__ __forceinline__ RefLight light_at(const SceneView& S, const FrameParams& F, int l) {
    RefLight Lg = F.lights[l & (RT_MAX_LIGHTS - 1)];
    if (l >= RT_MAX_LIGHTS) {
        const volatile uint32_t* p = reinterpret_cast<const volatile uint32_t*>(S.lights + l);
        uint32_t w[12];
#pragma unroll
        for (int k = 0; k < 12; k++) w[k] = p[k];
        Lg.loc.x = __uint_as_float(w[0]); Lg.loc.y = __uint_as_float(w[1]); Lg.loc.z = __uint_as_float(w[2]); Lg.strength = __uint_as_float(w[3]);
        Lg.dir.x = __uint_as_float(w[4]); Lg.dir.y = __uint_as_float(w[5]); Lg.dir.z = __uint_as_float(w[6]); Lg.lightType = w[7];
        Lg.color.x = __uint_as_float(w[8]); Lg.color.y = __uint_as_float(w[9]); Lg.color.z = __uint_as_float(w[10]); Lg.pad1 = __uint_as_float(w[11]);
    }
    return Lg;
}

__device__ __forceinline__ void slot_to_pixel(const FrameParams& F, uint32_t slot, int* px, int* py) {
    uint32_t tile = slot >> 6, l = slot & 63u;
    uint32_t tx = tile % (uint32_t)F.tilesX, ty = tile / (uint32_t)F.tilesX;
    *px = F.rx + (int)(tx * 8u + (l & 7u));
    *py = F.ry + (int)(ty * 8u + (l >> 3));
}

// ---------------------------------------------------------------------------------------------
// shader.comp:171-184 — same float sequence; also returns tEnter for ordering / culling.
// FAST: the hardware min/max (v_min_f32 / v_max3_f32 ...).  They equal the GLSL selects `y<x?y:x` / `x<y?y:x`
// whenever no operand is NaN (a +-0 pick cannot change the two comparisons below), and a NaN can only appear here
// as 0 * inf, i.e. for a ray with a zero / non-finite direction component or a non-finite origin.  Waves holding
// such a ray run the select form (ray_is_plain below), so both forms return identical results.
template <bool FAST>
__device__ __forceinline__ bool slab(v3 o, v3 inv, const float* lo, const float* hi, float* tEnterOut) {
    float t0x = (lo[0] - o.x) * inv.x, t0y = (lo[1] - o.y) * inv.y, t0z = (lo[2] - o.z) * inv.z;
    float t1x = (hi[0] - o.x) * inv.x, t1y = (hi[1] - o.y) * inv.y, t1z = (hi[2] - o.z) * inv.z;
    float tEnter, tExit;
    if (FAST) {
        tEnter = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
        tExit = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
    } else {
        float tminx = gmin(t0x, t1x), tminy = gmin(t0y, t1y), tminz = gmin(t0z, t1z);
        float tmaxx = gmax(t0x, t1x), tmaxy = gmax(t0y, t1y), tmaxz = gmax(t0z, t1z);
        tEnter = gmax(gmax(tminx, tminy), tminz);
        tExit = gmin(gmin(tmaxx, tmaxy), tmaxz);
    }
    *tEnterOut = tEnter;
    return tExit >= tEnter && tExit >= 0.0f;
}

// true when the ray can never produce a NaN in a slab test: finite origin, finite non-zero direction components
__device__ __forceinline__ bool ray_is_plain(v3 o, v3 d) {
    const float big = 3.0e38f;
    bool od = fabsf(o.x) < big && fabsf(o.y) < big && fabsf(o.z) < big;
    bool dd = fabsf(d.x) < big && fabsf(d.y) < big && fabsf(d.z) < big && d.x != 0.f && d.y != 0.f && d.z != 0.f;
    return od && dd;
}

// Ray origin and reciprocal direction as the register pairs the packed slab test consumes.
typedef float f2 __attribute__((ext_vector_type(2)));
struct RayPairs { f2 oxy, ozx, oyz, ixy, izx, iyz; };
__device__ __forceinline__ RayPairs make_pairs(v3 o, v3 inv) {
    RayPairs R;
    R.oxy = f2{o.x, o.y}; R.ozx = f2{o.z, o.x}; R.oyz = f2{o.y, o.z};
    R.ixy = f2{inv.x, inv.y}; R.izx = f2{inv.z, inv.x}; R.iyz = f2{inv.y, inv.z};
    return R;
}

// slab<true> on both child boxes of a CNode (n0..n2 = its first three 16-byte words): the same (bound - o) * inv
// per component, evaluated two at a time with packed FP32 subtract / multiply (v_pk_add_f32, v_pk_mul_f32 round each
// half exactly like the scalar instructions), so a node costs 12 packed instead of 24 scalar operations there.
__device__ __forceinline__ void slab_pair(const RayPairs& R, float4 n0, float4 n1, float4 n2,
                                          float& te0, float& tx0, float& te1, float& tx1) {
    f2 a = (f2{n0.x, n0.y} - R.oxy) * R.ixy;   // lo0.x lo0.y
    f2 b = (f2{n0.z, n0.w} - R.ozx) * R.izx;   // lo0.z hi0.x
    f2 c = (f2{n1.x, n1.y} - R.oyz) * R.iyz;   // hi0.y hi0.z
    f2 d = (f2{n1.z, n1.w} - R.oxy) * R.ixy;   // lo1.x lo1.y
    f2 e = (f2{n2.x, n2.y} - R.ozx) * R.izx;   // lo1.z hi1.x
    f2 f = (f2{n2.z, n2.w} - R.oyz) * R.iyz;   // hi1.y hi1.z
    te0 = fmaxf(fmaxf(fminf(a.x, b.y), fminf(a.y, c.x)), fminf(b.x, c.y));
    tx0 = fminf(fminf(fmaxf(a.x, b.y), fmaxf(a.y, c.x)), fmaxf(b.x, c.y));
    te1 = fmaxf(fmaxf(fminf(d.x, e.y), fminf(d.y, f.x)), fminf(e.x, f.y));
    tx1 = fminf(fminf(fmaxf(d.x, e.y), fmaxf(d.y, f.x)), fmaxf(e.x, f.y));
}

struct HitRec { float dist; float t; int idx; int ref; v3 loc; };

// shader.comp:104-125 + 269-276 on a pre-transformed triangle (A, B, C = the record's three 16-byte words).
template <bool EXACT>
__device__ __forceinline__ void test_tri(float4 A, float4 B, float4 C, int idx, v3 o, v3 d, HitRec& best) {
    // All three words of the record are fetched before the determinant test: it almost never rejects, and left to
    // itself the compiler sinks the load of v0 behind it, a second dependent round trip per triangle.
    asm volatile("" : "+v"(A.x), "+v"(A.y), "+v"(A.z));
    v3 v0 = mk(A.x, A.y, A.z), e1 = mk(A.w, B.x, B.y), e2 = mk(B.z, B.w, C.x);
    v3 pv = cross3(d, e2);
    float det = dot3(e1, pv);
    if (fabsf(det) < 1e-8f) return;
    float invDet = 1.0f / det;
    v3 s = o - v0;
    float u = dot3(s, pv) * invDet;
    if (u < 0.0f || u > 1.0f) return;
    v3 q = cross3(s, e1);
    float v = dot3(d, q) * invDet;
    if (v < 0.0f || u + v > 1.0f) return;
    float t = dot3(e2, q) * invDet;
    if (t < 1e-8f) return;
    v3 h = o + d * t;
    v3 oh = o - h;
    float newDist = dot3(oh, oh);
    // Reference: strict < in visit order.  Visit rank == array index, so (dist, idx) lexicographic
    // minimum is the same winner under any traversal order.
    bool better = EXACT ? (newDist < best.dist) : (newDist < best.dist || (newDist == best.dist && idx < best.idx));
    if (better) {
        best.dist = newDist; best.t = t; best.idx = idx; best.loc = h;
        best.ref = __float_as_int(C.y);
    }
}

// Per-lane traversal stack: the first RT_LDS_STACK entries live in LDS (entry k of thread t at lds[k*RT_BLOCK + t]:
// consecutive lanes hit consecutive 8-byte slots, conflict-free), deeper entries in a global overflow area sized
// from the tree depth at scene load (coalesced the same way).  No scratch memory is used.  A push beyond the
// total capacity (>= 64 entries = shader.comp:246) is dropped like the reference's `stackPtr < 64` guard.
struct Stack {
    typedef unsigned long long __attribute__((address_space(1))) GlobalEntry;      // an entry of the overflow area (x low, y high), named as global memory
    uint2* lds; GlobalEntry* ov; uint32_t ovStride; int cap;
    int ldsDepth, ldsStride;   // entries kept in LDS per lane, distance between a lane's consecutive entries (= workgroup size)
    // The overflow pointer carries the global address space so that the compiler keeps these accesses apart from the LDS
    // ones (merged through a generic pointer, both become flat_* instructions on a selected address, which are slower
    // than ds_* and count on two wait counters) — and each entry moves as one cached 8-byte access.  (They used to be
    // volatile dword pairs: two system-scope round trips per entry, which a tree deeper than the LDS part pays on every
    // deep push and pop.)
    __device__ __forceinline__ void push(int& sp, uint2 e) const {
        if (sp < ldsDepth) lds[sp * ldsStride] = e;
        else if (sp < cap) ov[(size_t)(sp - ldsDepth) * ovStride] = (unsigned long long)e.x | ((unsigned long long)e.y << 32);
        if (sp < cap) sp++;
    }
    __device__ __forceinline__ uint2 pop(int& sp) const {
        sp--;
        if (sp < ldsDepth) return lds[sp * ldsStride];
        const unsigned long long v = ov[(size_t)(sp - ldsDepth) * ovStride];
        uint2 e; e.x = (uint32_t)v; e.y = (uint32_t)(v >> 32);
        return e;
    }
};
// ldsBase: ldsDepth x block entries (block = workgroup size).  The overflow area holds P.overflowDepth entries per thread of the grid
// beyond the ldsDepth LDS ones (sized on the host for the smallest ldsDepth any kernel uses).
__device__ __forceinline__ Stack make_stack(uint2* ldsBase, const struct PathBuffers& P, int ldsDepth = RT_LDS_STACK, int block = RT_BLOCK) {
    Stack st;
    st.lds = ldsBase + threadIdx.x;
    st.ldsDepth = ldsDepth; st.ldsStride = block;
    st.ov = (Stack::GlobalEntry*)(P.overflow + (blockIdx.x * (uint32_t)block + threadIdx.x));
    st.ovStride = P.overflowStride;
    st.cap = ldsDepth + (int)P.overflowDepth;
    return st;
}

#define RT_DONE ((int)0x80000000)   // traversal finished (never a valid leaf code: codes are < 2^31)

// Closest hit == lineCast (shader.comp:243-290), "while-while" form: an inner loop that only descends internal
// nodes and a leaf step that only tests triangles, so a wave executes each body with the lanes that need it instead
// of both bodies per step.
//   EXACT:  reference order (right child first), no culling, leaf children always entered.
//   !EXACT: near child first, distance culling with slack, padded leaf boxes.  Same winner: DESIGN.md.
__device__ __forceinline__ bool first_active_lane() {
    unsigned long long m = __ballot(true);
    return (m & ((1ull << (threadIdx.x & 63u)) - 1ull)) == 0ull;
}

// wSteps (COUNT only, may be null): [0] += wave-level executions of the node step, [1] += of the triangle step,
// counted by the first active lane of each execution (lane utilisation = visits / (64 x executions)).
template <bool EXACT, bool COUNT, bool FAST>
__device__ __forceinline__ bool closest_hit(const SceneView& S, v3 o, v3 d, const Stack& stk, HitRec& best,
                                            uint32_t& nNodes, uint32_t& nTris, uint32_t* wSteps = nullptr) {
    best.dist = 3.40282347e+38f; best.t = __int_as_float(0x7f800000); best.idx = -1; best.ref = -1;
    best.loc = mk(0.f, 0.f, 0.f);
    v3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    float te;
    if (COUNT) nNodes++;
    if (!slab<FAST>(o, inv, S.rootLo, S.rootHi, &te)) return false;
    int sp = 0;
    int cur = S.rootChild;
    // EXACT: number of entries on the reference's own stack (shader.comp:246) while the current node is processed.  The
    // shader pushes both children of a node while there is room for them (64 entries) and box-tests a node when it pops
    // it; this walk pushes only siblings whose box passed, so it counts the others: a push the shader would drop for
    // lack of room is dropped here too, and trees deeper than 64 levels come out as the shader renders them.
    int refSp = 0;
    const RayPairs R = make_pairs(o, inv);
    auto popNext = [&]() {
        cur = RT_DONE;
        while (sp > 0) {
            uint2 e = stk.pop(sp);
            if (!EXACT) {
                float lim = best.t + best.t * 1e-4f + S.cullAbs;
                if (__uint_as_float(e.y) > lim) continue;
            } else refSp = (int)e.y;          // the sibling's position on the reference stack = entries left below it
            cur = (int)e.x;
            break;
        }
    };
    while (cur != RT_DONE) {
        while (cur >= 0) {
            const float4* np = reinterpret_cast<const float4*>(S.nodes + cur);
            float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
            if (COUNT) { nNodes++; if (wSteps && first_active_lane()) wSteps[0]++; }
            int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
            float t0, t1;
            bool p0, p1;
            if (FAST) {
                float x0, x1;
                slab_pair(R, n0, n1, n2, t0, x0, t1, x1);
                p0 = x0 >= t0 && x0 >= 0.0f;
                p1 = x1 >= t1 && x1 >= 0.0f;
            } else {
                float lo0[3] = {n0.x, n0.y, n0.z}, hi0[3] = {n0.w, n1.x, n1.y};
                float lo1[3] = {n1.z, n1.w, n2.x}, hi1[3] = {n2.y, n2.z, n2.w};
                p0 = slab<false>(o, inv, lo0, hi0, &t0);
                p1 = slab<false>(o, inv, lo1, hi1, &t1);
            }
            if (EXACT) {
                // a leaf child's own box is never tested by the reference (Q2): enter unconditionally;
                // the reference pops children.y first (shader.comp:281-284)
                p0 = p0 || (c0 < 0);
                p1 = p1 || (c1 < 0);
                // children.x, then children.y, each pushed if present and stackPtr < 64 (shader.comp:281-284); y is popped next
                const int pos0 = refSp;
                const bool in0 = c0 != -1 && refSp < 64;
                if (in0) refSp++;
                const bool in1 = c1 != -1 && refSp < 64;
                if (in0 && p0) { uint2 e; e.x = (uint32_t)c0; e.y = (uint32_t)pos0; stk.push(sp, e); }
                if (in1 && p1) cur = c1;
                else popNext();
            } else {
                float lim = best.t + best.t * 1e-4f + S.cullAbs;
                p0 = p0 && !(t0 > lim);
                p1 = p1 && !(t1 > lim);
                if (p0 && p1) {
                    bool swap = t1 < t0;
                    uint2 e; e.x = (uint32_t)(swap ? c0 : c1); e.y = __float_as_uint(swap ? t0 : t1);
                    stk.push(sp, e);
                    cur = swap ? c1 : c0;
                } else if (p0) cur = c0;
                else if (p1) cur = c1;
                else popNext();
            }
        }
        if (cur != RT_DONE) {
            int code = -cur - 1;
            int first = code >> 3, cnt = code & 7;
            const float4* tp = reinterpret_cast<const float4*>(S.tris + first);
            for (int k = 0; k < cnt; k++) {
                if (COUNT) { nTris++; if (wSteps && first_active_lane()) wSteps[1]++; }
                test_tri<EXACT>(tp[3 * k], tp[3 * k + 1], tp[3 * k + 2], first + k, o, d, best);
            }
            popNext();
        }
    }
    return best.idx >= 0;
}

// Any-hit for shadow rays.  kind 1 (sun): occluded on any hit (shader.comp:481-483).
// kind 0 (point): occluded when the closest hit H has distance(H, shadeLoc) < distToLight
// (shader.comp:467-471).  In traversalMode 0 the first hit satisfying that predicate ends the walk and
// boxes entered beyond distToLight (+slack) are skipped; traversalMode 1 runs the full closest-hit.
template <bool COUNT, bool FAST>
__device__ __forceinline__ bool occluded_fast(const SceneView& S, v3 o, v3 d, int kind, v3 shadeLoc, float distToLight,
                                              const Stack& stk, uint32_t& nNodes, uint32_t& nTris) {
    v3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    float te;
    if (COUNT) nNodes++;
    if (!slab<FAST>(o, inv, S.rootLo, S.rootHi, &te)) return false;
    // For point lights |d| = 1 so t ~ distance from o; o is within ~1e-5*|n| of shadeLoc.
    float lim = (kind == 0) ? (distToLight + distToLight * 1e-3f + S.cullAbs * 4.0f) : __int_as_float(0x7f800000);
    int sp = 0;
    int cur = S.rootChild;
    bool occ = false;
    const RayPairs R = make_pairs(o, inv);
    while (cur != RT_DONE) {
        while (cur >= 0) {
            const float4* np = reinterpret_cast<const float4*>(S.nodes + cur);
            float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
            if (COUNT) nNodes++;
            int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
            float t0, t1;
            bool p0, p1;
            if (FAST) {
                float x0, x1;
                slab_pair(R, n0, n1, n2, t0, x0, t1, x1);
                p0 = x0 >= t0 && x0 >= 0.0f && !(t0 > lim);
                p1 = x1 >= t1 && x1 >= 0.0f && !(t1 > lim);
            } else {
                float lo0[3] = {n0.x, n0.y, n0.z}, hi0[3] = {n0.w, n1.x, n1.y};
                float lo1[3] = {n1.z, n1.w, n2.x}, hi1[3] = {n2.y, n2.z, n2.w};
                p0 = slab<false>(o, inv, lo0, hi0, &t0) && !(t0 > lim);
                p1 = slab<false>(o, inv, lo1, hi1, &t1) && !(t1 > lim);
            }
            if (p0 && p1) {      // nearer box first: an occluder, if any, tends to be found sooner
                bool swap = t1 < t0;
                uint2 e; e.x = (uint32_t)(swap ? c0 : c1); e.y = 0u; stk.push(sp, e);
                cur = swap ? c1 : c0;
            }
            else if (p0) cur = c0;
            else if (p1) cur = c1;
            else if (sp > 0) cur = (int)stk.pop(sp).x;
            else cur = RT_DONE;
        }
        if (cur != RT_DONE) {
            int code = -cur - 1;
            int first = code >> 3, cnt = code & 7;
            const float4* tp = reinterpret_cast<const float4*>(S.tris + first);
            for (int k = 0; k < cnt; k++) {
                if (COUNT) nTris++;
                HitRec h; h.dist = 3.40282347e+38f; h.idx = -1; h.t = 0.f; h.ref = -1; h.loc = mk(0.f, 0.f, 0.f);
                test_tri<true>(tp[3 * k], tp[3 * k + 1], tp[3 * k + 2], first + k, o, d, h);
                if (h.idx >= 0) {
                    if (kind == 1) occ = true;
                    else { v3 dl = h.loc - shadeLoc; if (sqrtf(dot3(dl, dl)) < distToLight) occ = true; }
                }
            }
            if (occ) cur = RT_DONE;
            else if (sp > 0) cur = (int)stk.pop(sp).x;
            else cur = RT_DONE;
        }
    }
    return occ;
}

// Slot numbering of a pass.  A slot is one (sample of the pass, pixel) pair; bounce 0 has no queue (entry i of its launches
// is slot i), and later queues keep that order, so the numbering decides which rays share a wave.  A group of 64
// consecutive slots holds 2^grpShift consecutive samples of 64 >> grpShift neighbouring pixels (pixel slots in tile order,
// slot_to_pixel): with grpShift = 0 a wave traces one sample of an 8 x 8 pixel tile, with grpShift = 6 sixty-four samples of
// one pixel — rays that start together and, for the camera and the sun's shadow rays, run side by side through the tree.
// Groups are ordered sample block first, pixel group second.  (The pass's sample count is a multiple of 2^grpShift and
// slotsPerSample a multiple of 64: rt_core.hip.)
__device__ __forceinline__ void slot_split(const PathBuffers& P, uint32_t slot, uint32_t* sample, uint32_t* pixelSlot) {
    const uint32_t sh = P.grpShift, psh = 6u - sh;
    const uint32_t l = slot & 63u, G = slot >> 6;
    const uint32_t nPG = P.slotsPerSample >> psh;            // pixel groups per sample block
    const uint32_t sb = G / nPG, pg = G - sb * nPG;
    *sample = (sb << sh) + (l >> psh);
    *pixelSlot = (pg << psh) + (l & ((1u << psh) - 1u));
}
__device__ __forceinline__ uint32_t slot_of(const PathBuffers& P, uint32_t sample, uint32_t pixelSlot) {
    const uint32_t sh = P.grpShift, psh = 6u - sh;
    const uint32_t nPG = P.slotsPerSample >> psh;
    return ((((sample >> sh) * nPG + (pixelSlot >> psh)) << 6) | ((sample & ((1u << sh) - 1u)) << psh) | (pixelSlot & ((1u << psh) - 1u)));
}

// ---------------------------------------------------------------------------------------------
// shader.comp:309-324: the camera ray of slot `slot` (sample rBase + slot / slotsPerSample of its pixel).  Bounce 0
// needs neither a queue nor stored rays: every slot carries a path and k_extend / k_shade regenerate the ray from
// the slot index (a deterministic function, so both see identical bits).  Returns false for slots of edge tiles
// that fall outside the region.
#define RT_FLAG_REFRACTING 1u
__device__ __forceinline__ bool primary_ray(const FrameParams& F, const PathBuffers& P, int rBase, uint32_t slot,
                                            int* pxOut, int* pyOut, int* rOut, v3* dirOut) {
    int px, py;
    uint32_t sm, ps;
    slot_split(P, slot, &sm, &ps);
    int r = rBase + (int)sm;
    slot_to_pixel(F, ps, &px, &py);
    *pxOut = px; *pyOut = py; *rOut = r;
    int N = F.gridN;
    int gx = r % N, gy = r / N;
    float jx = ((float)gx + 0.5f) / (float)N, jy = ((float)gy + 0.5f) / (float)N;
    float ux = ((float)px + jx) / (float)F.width * 2.0f - 1.0f;
    float uy = ((float)py + jy) / (float)F.height * 2.0f - 1.0f;
    ux *= F.aspect;
    float fx = ux * F.tanHalfFov;
    float fy = uy * F.tanHalfFov;
    v3 fwd = ld3(F.camForward), right = ld3(F.camRight), up = ld3(F.camUp);
    *dirOut = (fwd + fx * right + fy * up);
    return px < F.rx + F.rw && py < F.ry + F.rh;
}

// rays: the bounce's ray buffer (PRIMARY: unused, entry i is slot i).  rayCount > 0 overrides the number of rays added to the counters
// (bounce 0: region pixels, not slots).
// The traversal kernels below are written as bodies (…_batch, …_refill) that take their block index and grid size as
// arguments, with a __global__ wrapper each: k_trace_pair runs two of them in one launch (rt_trace_refill.hip.h).
// lds: RT_LDS_STACK x RT_BLOCK stack entries of the workgroup.
template <bool EXACT, bool COUNT, bool PRIMARY>
__device__ __forceinline__ void extend_batch(const SceneView& S, const FrameParams& F, const PathBuffers& P, const float4* __restrict__ rays,
                                             const uint32_t* __restrict__ count, int rBase, uint32_t rayCount,
                                             DeviceStats* stats, uint2* lds, uint32_t bid, uint32_t nblk) {
    const Stack stk = make_stack(lds, P);
    uint32_t n = *count;
    uint32_t nNodes = 0, nTris = 0, maxSteps = 0;
    uint32_t wSteps[2] = {0u, 0u};
    for (uint32_t i = bid * RT_BLOCK + threadIdx.x; i < n; i += nblk * RT_BLOCK) {
        v3 ro, rd;
        if (PRIMARY) {
            int px, py, r;
            ro = ld3(F.camPos);
            if (!primary_ray(F, P, rBase, i, &px, &py, &r, &rd)) { P.hit[i] = make_float2(0.f, __int_as_float(-2)); continue; }
        } else {
            uint32_t fl, slot;
            unpack_ray(rays[2u * i], rays[2u * i + 1u], ro, rd, fl, slot);
        }
        HitRec best;
        uint32_t before = nNodes + nTris;
        if (__all(ray_is_plain(ro, rd))) closest_hit<EXACT, COUNT, true>(S, ro, rd, stk, best, nNodes, nTris, wSteps);
        else closest_hit<EXACT, COUNT, false>(S, ro, rd, stk, best, nNodes, nTris, wSteps);
        if (COUNT) { uint32_t steps = nNodes + nTris - before; if (steps > maxSteps) maxSteps = steps; }
        P.hit[i] = make_float2(best.t, __int_as_float(best.ref));
    }
    if (bid == 0 && threadIdx.x == 0)
        atomicAdd(PRIMARY ? &stats->rays_primary : &stats->rays_bounce, (unsigned long long)(rayCount ? rayCount : n));
    if (COUNT) {
        __shared__ uint32_t red[4];
        if (threadIdx.x < 4) red[threadIdx.x] = 0;
        __syncthreads();
        for (int off = 32; off > 0; off >>= 1) {
            nNodes += __shfl_down(nNodes, off, 64); nTris += __shfl_down(nTris, off, 64);
            wSteps[0] += __shfl_down(wSteps[0], off, 64); wSteps[1] += __shfl_down(wSteps[1], off, 64);
        }
        for (int off = 32; off > 0; off >>= 1) { uint32_t o2 = __shfl_down(maxSteps, off, 64); if (o2 > maxSteps) maxSteps = o2; }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&red[0], nNodes); atomicAdd(&red[1], nTris); atomicAdd(&red[2], wSteps[0]); atomicAdd(&red[3], wSteps[1]);
            if (maxSteps) atomicMax(&stats->max_ray_steps, (unsigned long long)maxSteps);
        }
        __syncthreads();
        if (threadIdx.x == 0 && (red[0] | red[1])) {
            atomicAdd(&stats->node_visits, (unsigned long long)red[0]);
            atomicAdd(&stats->tri_tests, (unsigned long long)red[1]);
            if (!PRIMARY) {
                atomicAdd(&stats->bounce_node_visits, (unsigned long long)red[0]);
                atomicAdd(&stats->bounce_tri_tests, (unsigned long long)red[1]);
                atomicAdd(&stats->wave_node_steps, (unsigned long long)red[2]);
                atomicAdd(&stats->wave_tri_steps, (unsigned long long)red[3]);
            }
        }
    }
}

template <bool EXACT, bool COUNT, bool PRIMARY>
__global__ __launch_bounds__(RT_BLOCK) void k_extend(SceneView S, FrameParams F, PathBuffers P, const float4* __restrict__ rays,
                                                     const uint32_t* __restrict__ count, int rBase, uint32_t rayCount,
                                                     DeviceStats* stats) {
    __shared__ uint2 lds[RT_LDS_STACK * RT_BLOCK];
    extend_batch<EXACT, COUNT, PRIMARY>(S, F, P, rays, count, rBase, rayCount, stats, lds, blockIdx.x, gridDim.x);
}

// texture(sampler2D(handle), uv): LINEAR, REPEAT, level 0, RGBA8 unorm (Texture.cpp:5-34; GL 4.3 8.14).
__device__ __forceinline__ float4 sample_tex(const SceneView& S, uint64_t handle, float u, float v) {
    if (handle == 0 || handle > S.n_textures) return make_float4(0.f, 0.f, 0.f, 1.f);
    TexDesc T = S.textures[handle - 1];
    float fu = u * (float)T.width - 0.5f, fv = v * (float)T.height - 0.5f;
    float iu = floorf(fu), iv = floorf(fv);
    float a = fu - iu, b = fv - iv;
    long long w = T.width, h = T.height;
    long long i0 = (long long)iu % w; if (i0 < 0) i0 += w;
    long long j0 = (long long)iv % h; if (j0 < 0) j0 += h;
    long long i1 = (i0 + 1) % w, j1 = (j0 + 1) % h;
    const uchar4* px = reinterpret_cast<const uchar4*>(T.rgba8);
    uchar4 q00 = px[j0 * w + i0], q10 = px[j0 * w + i1], q01 = px[j1 * w + i0], q11 = px[j1 * w + i1];
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float4 o;
    o.x = w00 * ((float)q00.x / 255.0f) + w10 * ((float)q10.x / 255.0f) + w01 * ((float)q01.x / 255.0f) + w11 * ((float)q11.x / 255.0f);
    o.y = w00 * ((float)q00.y / 255.0f) + w10 * ((float)q10.y / 255.0f) + w01 * ((float)q01.y / 255.0f) + w11 * ((float)q11.y / 255.0f);
    o.z = w00 * ((float)q00.z / 255.0f) + w10 * ((float)q10.z / 255.0f) + w01 * ((float)q01.z / 255.0f) + w11 * ((float)q11.z / 255.0f);
    o.w = w00 * ((float)q00.w / 255.0f) + w10 * ((float)q10.w / 255.0f) + w01 * ((float)q01.w / 255.0f) + w11 * ((float)q11.w / 255.0f);
    return o;
}

// shader.comp:193-241
__device__ __forceinline__ v3 sample_ggx(v3 N, float rand1, float rand2, float alpha) {
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
    portable_sincos(phi, &sp, &cp);
    v3 Ht = mk(sinTheta * cp, sinTheta * sp, cosTheta);
    v3 T;
    if (fabsf(N.z) < 0.999f) T = normalize3(cross3(mk(0.f, 0.f, 1.f), N));
    else T = normalize3(cross3(mk(1.f, 0.f, 0.f), N));
    v3 B = cross3(N, T);
    v3 H = T * Ht.x + B * Ht.y + N * Ht.z;
    float lenH = length3(H);
    if (lenH < 1e-6f) return N;
    return H / lenH;
}

// shader.comp:329-452 and 524-531 for ONE path whose closest hit is h2 = (t, reference triangle id; -1 miss, -2 dead slot):
// miss / ambient, surface fetch, dielectric / GGX bounce, the light loop's contributions into the log (entries that await a
// shadow ray get w = 1).  In: the ray that was traced (rayLoc, rayDir), the path's strength and refraction flag.  Out: the
// next ray and strength, cont (the path goes on to bounce + 1), wantShadow (a log entry awaits its shadow ray, whose origin
// is the new rayLoc).  hitLoc = rayLoc + rayDir * t, formed by the caller.
// shader.comp:331-336: what a path that leaves the scene adds, as a log entry (w = 2: add).
__device__ __forceinline__ LogEntry miss_entry(int hasSun, v3 strength) {
    v3 ambient = hasSun ? mk(0.4431372549f, 0.73725490196f, 1.f) : mk(0.f, 0.f, 0.f);
    v3 cc = ambient * strength;
    return log_entry(cc.x, cc.y, cc.z);
}
#define RT_DEPTH_MISS0 0x80000000u      // depth word of a slot whose camera ray missed: 1 | this, and no log entry (k_resolve adds miss_entry)
struct ShadeResult { v3 o, d, s; bool refracting, cont, wantShadow; uint32_t pending, skipped; };
template <bool TEX, bool GLASS, bool MANY, bool SUN1>
__device__ __forceinline__ ShadeResult shade_path(const SceneView& S, const FrameParams& F, const PathBuffers& P, int bounce, int px, int py, int r,
                                                  uint32_t logIdx, float2 h2, v3 rayLoc, v3 rayDir, v3 strength, bool refracting, const v3 hitLoc,
                                                  float4* imgAlbedo, float4* imgNormal, float4* imgDepth, int32_t* primTri, float* primHit) {
    bool cont = false, wantShadow = false;
    uint32_t pending = 0, skipped = 0;
    int ref = __float_as_int(h2.y);
    v3 camPos = ld3(F.camPos);
    const LogPlane LP = log_plane(P, bounce);
    LogEntry* logp = LP.base + logIdx;                 // entry of light l at logp[l * LP.stride]
    {      // (only hits get here: misses and dead slots are k_shade's fast path)
        const float4* sp = reinterpret_cast<const float4*>(S.shade + ref);
        const float4 q0 = sp[0], q1 = sp[1], q2 = sp[2], q3 = sp[3], q4 = sp[4];
        v3 a = mk(q0.x, q0.y, q0.z), b = mk(q0.w, q1.x, q1.y), c = mk(q1.z, q1.w, q2.x);
        v3 n1 = mk(q2.y, q2.z, q2.w), n2 = mk(q3.x, q3.y, q3.z), n3 = mk(q3.w, q4.x, q4.y);
        uint32_t matIdx = __float_as_uint(q4.z);
        const float4* m4 = reinterpret_cast<const float4*>(S.cmat + matIdx);
        float4 m0 = m4[0], m1 = m4[1];
        const uint32_t mflags = TEX ? __float_as_uint(m1.w) : 0u;
        const bool hasBase = (mflags & 1u) != 0u, hasNormal = (mflags & 2u) != 0u, hasMR = (mflags & 4u) != 0u;
        float uv1x = 0.f, uv1y = 0.f, uv2x = 0.f, uv2y = 0.f, uv3x = 0.f, uv3y = 0.f;
        float4 q6 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mflags != 0u) {          // UVs and vertex ids only matter to textured / normal-mapped materials
            const float4* up = reinterpret_cast<const float4*>(S.shadeUV + ref);
            float4 u0 = up[0], u1 = up[1], u2 = up[2];
            uv1x = u0.x; uv1y = u0.y; uv2x = u0.z; uv2y = u0.w; uv3x = u1.x; uv3y = u1.y;
            q6 = make_float4(0.f, u1.z, u1.w, u2.x);
        }
        // barycentric, shader.comp:132-151
        v3 v0 = b - a, v1 = c - a, v2 = hitLoc - a;
        float d00 = dot3(v0, v0), d01 = dot3(v0, v1), d11 = dot3(v1, v1), d20 = dot3(v2, v0), d21 = dot3(v2, v1);
        float denomB = d00 * d11 - d01 * d01;
        float bv = (d11 * d20 - d01 * d21) / denomB;
        float bw = (d00 * d21 - d01 * d20) / denomB;
        float bu = 1.0f - bv - bw;
        v3 hitNormal = (n1 * bu + n2 * bv + n3 * bw);
        float hitU = uv1x * bu + uv2x * bv + uv3x * bw;
        float hitV = uv1y * bu + uv2y * bv + uv3y * bw;
        const RefMaterial* mp = S.materials + matIdx;        // texture handles (read only under a flag)
        float transmission = m0.w, ior = m1.z;
        float metallicFactor, roughnessFactor;
        if (hasMR) {
            float4 t = sample_tex(S, mp->metallicRoughnessTexture, hitU, hitV);
            metallicFactor = t.x; roughnessFactor = t.y;
        } else { metallicFactor = m1.x; roughnessFactor = m1.y; }
        v3 normal;
        if (hasNormal) {
            uint32_t i1 = __float_as_uint(q6.y), i2 = __float_as_uint(q6.z), i3 = __float_as_uint(q6.w);
            const float* t1 = S.vertices[i1].tangent; const float* t2 = S.vertices[i2].tangent; const float* t3 = S.vertices[i3].tangent;
            v3 tangent = (ld3(t1) * bu + ld3(t2) * bv + ld3(t3) * bw);
            float handed = t1[3] * bu + t2[3] * bv + t3[3] * bw;
            float4 nt = sample_tex(S, mp->normalTexture, hitU, hitV);
            v3 normalTex = mk(nt.x, nt.y, nt.z) * 2.0f - mk(1.f, 1.f, 1.f);
            float normalStrength = 1.f;
            normalTex.x *= normalStrength; normalTex.y *= normalStrength;
            v3 bitangent = cross3(hitNormal, tangent) * handed;
            normal = tangent * normalTex.x + bitangent * normalTex.y + hitNormal * normalTex.z;
        } else normal = hitNormal;

        // reflect / refract, shader.comp:389-431
        bool shouldReflect = true;
        float rb = (float)(r * bounce);
        if (GLASS && transmission > 0) {
            float iorAir = 1.f;
            float cosi = gclamp(dot3(-rayDir, normal), -1.0f, 1.0f);
            float r0 = (iorAir - ior) / (iorAir + ior);
            r0 = r0 * r0;
            float reflectionStrength = r0 + (1.0f - r0) * pow5(1.0f - cosi);
            if (!refracting) {
                float rnd = hash13(mk((float)px * 2.3f, (float)py * 2.3f, rb * 18.f) * rayDir + rayLoc);
                if (rnd > reflectionStrength) {
                    rayDir = refract3(rayDir, normal, iorAir / ior);
                    rayLoc = hitLoc - normal * 1e-5f;
                    strength = strength * transmission;
                    refracting = true;
                    shouldReflect = false;
                }
            } else {
                v3 newLoc = hitLoc + normal * 1e-5f;
                v3 newDir = refract3(rayDir, -normal, ior / iorAir);
                if (length3(newDir) == 0.f) {
                    rayDir = reflect3(rayDir, -normal);
                    rayLoc = hitLoc - normal * 1e-5f;
                    refracting = true;
                } else {
                    rayLoc = newLoc;
                    rayDir = newDir;
                    refracting = false;
                }
                shouldReflect = false;
            }
        }
        if (shouldReflect) {
            v3 H = normal;                       // sampleGGX returns N for alpha == 0 (shader.comp:194): no hashes needed
            if (roughnessFactor != 0.f) {
                float h1 = hash13(mk((float)px, (float)py, rb) * rayDir + rayLoc);
                float h2 = hash13(mk((float)px * 37.f, (float)py * 37.f, rb * 37.f) * rayDir + rayLoc);
                H = sample_ggx(normal, h1, h2, roughnessFactor);
            }
            rayDir = reflect3(rayDir, H);
            rayLoc = hitLoc + (normal * 1e-5f);
        }

        // lighting set-up, shader.comp:433-451
        v3 baseColor = mk(m0.x, m0.y, m0.z);
        if (hasBase) {
            float4 t = sample_tex(S, mp->baseColorTexture, hitU, hitV);
            baseColor = mk(t.x, t.y, t.z);
        }
        v3 viewDir = normalize3(camPos - hitLoc);
        float dielectricF0 = .04f;
        if (GLASS && transmission > 0) { float e = (ior - 1.0f) / (ior + 1.0f); dielectricF0 = e * e; }
        v3 F0 = mk(dielectricF0, dielectricF0, dielectricF0) * (1.0f - metallicFactor) + baseColor * metallicFactor;
        float alpha = roughnessFactor * roughnessFactor;
        const float PI = 3.14159265359f;
        v3 diffuseColor = baseColor * (1.0f - metallicFactor);

        if (r == 0 && bounce == 0) {
            size_t pix = (size_t)py * F.width + px;
            if (imgAlbedo) {
                imgAlbedo[pix] = make_float4(diffuseColor.x, diffuseColor.y, diffuseColor.z, 1.f);
                imgNormal[pix] = make_float4(normal.x / 2.f + 0.5f, normal.y / 2.f + 0.5f, normal.z / 2.f + 0.5f, 1.f);
                float dd = length3(camPos - hitLoc) / 255.f;
                imgDepth[pix] = make_float4(dd, dd, dd, 1.f);
            }
            if (primTri) {
                primTri[pix] = ref;
                primHit[pix * 3] = hitLoc.x; primHit[pix * 3 + 1] = hitLoc.y; primHit[pix * 3 + 2] = hitLoc.z;
            }
        }

        // light loop, shader.comp:453-522 minus the shadow lineCast (k_shadow)
        if (!SUN1 && F.numLights == 0) logp[0] = log_entry(0.f, 0.f, 0.f);
        const int nLightsHere = SUN1 ? 1 : F.numLights;
        for (int l = 0; l < nLightsHere; l++) {
            const RefLight Lg = MANY ? light_at(S, F, l) : F.lights[l];
            float4 L0 = make_float4(Lg.loc.x, Lg.loc.y, Lg.loc.z, Lg.strength), L1 = make_float4(Lg.dir.x, Lg.dir.y, Lg.dir.z, __uint_as_float(Lg.lightType)), L2 = make_float4(Lg.color.x, Lg.color.y, Lg.color.z, 0.f);
            const uint32_t ltype = SUN1 ? 1u : __float_as_uint(L1.w);
            v3 dirToLight = mk(0.f, 0.f, 0.f);
            float attenuation = 1.f;
            bool cast = false;
            if (ltype == 0u) {
                v3 toL = mk(L0.x, L0.y, L0.z) - hitLoc;
                dirToLight = normalize3(toL);
                float distToLight = length3(toL);
                attenuation = L0.w / (1.f + 1.f * distToLight + 1.f * distToLight * distToLight);
                cast = true;
            } else if (ltype == 1u) {
                dirToLight = -mk(L1.x, L1.y, L1.z);
                cast = true;
            }
            float NdotL = gmax(dot3(normal, dirToLight), 0.0f);
            LogEntry pendOut = log_entry(0.f, 0.f, 0.f);
            if (!(NdotL <= 0.0f)) {
                v3 Hh = normalize3(viewDir + dirToLight);
                float NdotH = gmax(dot3(normal, Hh), 0.0f);
                float NdotV = gmax(dot3(normal, viewDir), 0.0f);
                float cosTheta = gclamp(dot3(normal, -viewDir), 0.0f, 1.0f);
                v3 Fr = F0 + (1.0f - F0) * pow5(1.0f - cosTheta);
                float a2 = alpha * alpha;
                float denom = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;
                float D = a2 / (PI * denom * denom);
                float k = (alpha + 1) * (alpha + 1) / 8.0f;
                float G_V = NdotV / (NdotV * (1.0f - k) + k);
                float G_L = NdotL / (NdotL * (1.0f - k) + k);
                float G = G_V * G_L;
                v3 specular = (D * Fr * G) / (4.0f * NdotV * NdotL + 0.001f);
                v3 diffuse = diffuseColor / PI;
                v3 radiance = mk(L2.x, L2.y, L2.z) * attenuation;
                v3 contrib = (diffuse + specular) * radiance * NdotL * strength;
                // A contribution of exactly +-0 in all three channels (mirror metals: diffuse 0 and D = 0) leaves
                // the pixel sum unchanged bit for bit whether or not it is added (the sum starts at +0 and
                // x + (+-0) == x), so the shadow ray that would decide it is not traced.  NaN fails the test.
                const bool nothing = cast && contrib.x == 0.f && contrib.y == 0.f && contrib.z == 0.f;
                if (nothing) skipped++;
                else {
                    pendOut = log_entry(contrib.x, contrib.y, contrib.z);
                    if (cast) { wantShadow = true; pending++; }
                }
                // cast == false: a light type the shader does not know gets no shadow test (the shadow kernels skip such
                // lights) and its contribution (only reachable with a NaN normal) is added as it stands
            } else if (cast) skipped++;
            logp[(size_t)l * LP.stride] = pendOut;
        }

        // shader.comp:524-530
        if (transmission == 0.f)
            strength = strength * mk(gclamp(baseColor.x, 0.f, 1.f), gclamp(baseColor.y, 0.f, 1.f), gclamp(baseColor.z, 0.f, 1.f));
        bool active = !(length3(strength) < F.strengthCutoff);
        cont = active && (bounce + 1 < F.maxBounces);
    }
    ShadeResult R;
    R.o = rayLoc; R.d = rayDir; R.s = strength; R.refracting = refracting; R.cont = cont; R.wantShadow = wantShadow;
    R.pending = pending; R.skipped = skipped;
    return R;
}

// shader.comp:329-452 and 524-531 for one path; light loop body 486-521 is evaluated here and its
// result parked in `pend` until k_shadow has traced the shadow ray (453-484).
// TEX / GLASS: the scene has a textured (or normal-mapped) material / a transmissive one.  Scenes without them run a
// variant with those branches compiled out: fewer live registers on the common path.  MANY: more lights than travel in
// the kernel arguments (RT_MAX_LIGHTS); otherwise every light is a scalar kernarg read.  SUN1: exactly one light and it is
// a sun (every BASELINE config): no light loop, no point-light branch.
// LATE: the log has a compact part (PathBuffers::late) and this launch may be one of its bounces; without it that code is
// compiled out (its few uniform values cost the common variants a wave per SIMD: 135 instead of 125 registers).
template <bool TEX, bool GLASS, bool MANY, bool SUN1, bool LATE = false>
__global__ __launch_bounds__(RT_BLOCK, (TEX ? 3 : 4)) void k_shade(SceneView S, FrameParams F, PathBuffers P,
                                                    const uint32_t* __restrict__ count, uint32_t* countOut,
                                                    uint32_t* shadowCountOut, int rBase, int bounce,
                                                    float4* imgAlbedo, float4* imgNormal, float4* imgDepth,
                                                    int32_t* primTri, float* primHit, DeviceStats* stats) {
    __shared__ uint32_t ldsAppend[16];
    __shared__ uint32_t ldsCount[2];
    // The outgoing ray of an item waits here until the block knows the item's position in the next queue
    // (entry k of thread t at [k * RT_BLOCK + t]: written and read back by the same thread, no barrier needed).
    // 36 bytes per item: (o.xyz, d.x) (d.y, d.z, s.x, s.y) (s.z); slot and flag stay in registers.
    __shared__ float4 stageA[RT_ITEMS * RT_BLOCK], stageB[RT_ITEMS * RT_BLOCK];
    __shared__ float stageC[RT_ITEMS * RT_BLOCK];
    // PF: the chunk's four hit records are fetched up front (hit -> shading record -> material is a chain of dependent
    // round trips).  8 registers; the variants without textures still fit 128 (4 waves per SIMD) with it.  Fetching the
    // next item's shading record or ray state ahead as well costs a wave per SIMD and is slower (measured).
    constexpr bool PF = true;
    if (threadIdx.x < 2) ldsCount[threadIdx.x] = 0;
    const float4* __restrict__ raysIn = P.ray[bounce & 1];
    const float4* __restrict__ raySIn = P.rayS[bounce & 1];
    float4* __restrict__ raysOut = P.ray[(bounce + 1) & 1];
    float4* __restrict__ raySOut = P.rayS[(bounce + 1) & 1];
    uint32_t n = *count;
    // a bounce beyond the dense part of the log: entries at queue positions, of which the log holds lateCap.  A queue longer
    // than that is cut (the host sees the flag, discards the frame and renders it again with a dense log).
    const bool lateB = LATE && (uint32_t)bounce >= P.denseBounces;
    if (lateB && n > P.lateCap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) stats->late_overflow = 1ull;
        n = P.lateCap;
    }
    uint32_t* const lateNext = lateB ? P.lateNext + (size_t)((uint32_t)bounce - P.denseBounces) * P.lateCap : nullptr;
    const uint32_t chunk = RT_BLOCK * RT_ITEMS;
    uint32_t nChunks = (n + chunk - 1) / chunk;
    uint32_t skipped = 0, pending = 0;
    for (uint32_t ch = blockIdx.x; ch < nChunks; ch += gridDim.x) {
      uint32_t contMask = 0, shadowMask = 0, refrMask = 0;
      uint32_t slotK[RT_ITEMS] = {};
      const uint32_t iBase = ch * chunk + threadIdx.x;
      const float2 noHit = make_float2(0.f, __int_as_float(-2));
      float2 hq0 = noHit, hq1 = noHit, hq2 = noHit, hq3 = noHit;
      static_assert(RT_ITEMS >= 1 && RT_ITEMS <= 4, "hit-record queue below holds up to four items");
      if (PF) {
          if (iBase < n) hq0 = P.hit[iBase];
          if (RT_ITEMS > 1 && iBase + RT_BLOCK < n) hq1 = P.hit[iBase + RT_BLOCK];
          if (RT_ITEMS > 2 && iBase + 2 * RT_BLOCK < n) hq2 = P.hit[iBase + 2 * RT_BLOCK];
          if (RT_ITEMS > 3 && iBase + 3 * RT_BLOCK < n) hq3 = P.hit[iBase + 3 * RT_BLOCK];
      }
#pragma unroll 1
      for (int item = 0; item < RT_ITEMS; item++) {
        uint32_t i = iBase + (uint32_t)item * RT_BLOCK;
        bool live = i < n;
        bool cont = false, wantShadow = false;
        uint32_t slot = i;                                       // bounce 0: entry i is slot i
        float2 h2 = hq0;
        if (PF) { hq0 = hq1; hq1 = hq2; hq2 = hq3; hq3 = noHit; }
        else if (live) h2 = P.hit[i];
        const int ref = __float_as_int(h2.y);
        // A path that left the scene (or a dead slot) needs next to nothing — and on the benchmark scenes more than half of
        // the paths of every bounce are misses.  Bounce 0: the camera ray is not regenerated at all (the depth word alone says
        // "sky", k_resolve adds the constant; the primary-hit tap was preset to -1).  Later bounces: one 16-byte record, strength
        // and slot, is read; the ray's origin / direction and the pixel decode (three integer divisions) are not.
        if (live && ref < 0) {
            if (bounce == 0) P.depth[slot] = ref == -2 ? 0u : (1u | RT_DEPTH_MISS0);
            else {
                const float4 s4 = raySIn[i];
                slot = __float_as_uint(s4.w);
                const uint32_t logIdx = lateB ? i : slot;
                if (LATE && (uint32_t)bounce == P.denseBounces) P.lateHead[slot] = i;
                if (ref != -2) {
                    const LogPlane LP = log_plane(P, bounce);
                    LogEntry* logp = LP.base + logIdx;
                    logp[0] = miss_entry(F.hasSun, mk(s4.x, s4.y, s4.z));      // shader.comp:331-336
                    for (uint32_t l = 1; l < P.lightSlots; l++) logp[(size_t)l * LP.stride] = log_entry(0.f, 0.f, 0.f);
                }
                P.depth[slot] = ref == -2 ? 0u : (uint32_t)bounce + 1u;
            }
        } else if (live) {
            v3 rayLoc, rayDir, strength;
            bool refracting = false;
            int px, py, r;
            if (bounce == 0) {
                // camera ray regenerated from the slot index (see primary_ray)
                primary_ray(F, P, rBase, slot, &px, &py, &r, &rayDir);
                rayLoc = ld3(F.camPos);
                strength = mk(F.initialStrength, F.initialStrength, F.initialStrength);
            } else {
                float4 s4 = raySIn[i];
                uint32_t fl;
                unpack_ray(raysIn[2u * i], raysIn[2u * i + 1u], rayLoc, rayDir, fl, slot);
                strength = mk(s4.x, s4.y, s4.z);
                refracting = (fl & RT_FLAG_REFRACTING) != 0u;
                uint32_t sm, ps;
                slot_split(P, slot, &sm, &ps);
                r = rBase + (int)sm;
                slot_to_pixel(F, ps, &px, &py);
            }
            const uint32_t logIdx = lateB ? i : slot;
            if (LATE && (uint32_t)bounce == P.denseBounces) P.lateHead[slot] = i;
            const v3 hitLoc = rayLoc + rayDir * h2.x;      // == the h = o + d * t of the triangle test (shader.comp:122)
            const ShadeResult R = shade_path<TEX, GLASS, MANY, SUN1>(S, F, P, bounce, px, py, r, logIdx, h2, rayLoc, rayDir, strength, refracting, hitLoc,
                                                                     imgAlbedo, imgNormal, imgDepth, primTri, primHit);
            rayLoc = R.o; rayDir = R.d; strength = R.s; refracting = R.refracting; cont = R.cont; wantShadow = R.wantShadow;
            pending += R.pending; skipped += R.skipped;
            if (cont || wantShadow) {
                const uint32_t st = (uint32_t)item * RT_BLOCK + threadIdx.x;
                stageA[st] = make_float4(rayLoc.x, rayLoc.y, rayLoc.z, rayDir.x);
                stageB[st] = make_float4(rayDir.y, rayDir.z, strength.x, strength.y);
                stageC[st] = strength.z;
                if (refracting) refrMask |= 1u << item;
#pragma unroll
                for (int k = 0; k < RT_ITEMS; k++) if (item == k) slotK[k] = slot;
            }
            if (wantShadow && F.anyPoint) P.shadeLoc[logIdx] = make_float4(hitLoc.x, hitLoc.y, hitLoc.z, 0.f);
            // bounces shaded by this slot's path: written once, by the bounce at which the path ends
            if (!cont) P.depth[slot] = (uint32_t)bounce + 1u;
        }
        contMask |= (cont ? 1u : 0u) << item;
        shadowMask |= (wantShadow ? 1u : 0u) << item;
      }
      // compaction: next bounce's path queue and this bounce's shadow queue, one atomic each per block
      bool flag[2][RT_ITEMS];
      uint32_t idx[2][RT_ITEMS];
#pragma unroll
      for (int k = 0; k < RT_ITEMS; k++) { flag[0][k] = (contMask >> k) & 1u; flag[1][k] = (shadowMask >> k) & 1u; }
      uint32_t* const ctr[2] = {countOut, shadowCountOut};
      block_append<2>(flag, ctr, idx, ldsAppend);
#pragma unroll
      for (int k = 0; k < RT_ITEMS; k++) {
          const uint32_t ik = iBase + (uint32_t)k * RT_BLOCK;
          if (lateB && ik < n) lateNext[ik] = flag[0][k] ? idx[0][k] : 0xffffffffu;
          if (flag[0][k] | flag[1][k]) {
              const uint32_t st = (uint32_t)k * RT_BLOCK + threadIdx.x;
              const float4 a = stageA[st], b = stageB[st];
              const float sl = __uint_as_float(slotK[k]);
              if (flag[0][k]) {
                  const uint32_t o = idx[0][k];
                  raysOut[2u * o] = a;
                  raysOut[2u * o + 1u] = make_float4(b.x, b.y, __uint_as_float(((refrMask >> k) & 1u) ? RT_FLAG_REFRACTING : 0u), sl);
                  raySOut[o] = make_float4(b.z, b.w, stageC[st], sl);
              }
              if (flag[1][k]) P.shadowRec[idx[1][k]] = make_float4(a.x, a.y, a.z, lateB ? __uint_as_float(ik) : sl);
          }
      }
    }
    // ray counters: wave shuffle -> LDS -> one global atomic per block
    for (int off = 32; off > 0; off >>= 1) { skipped += __shfl_down(skipped, off, 64); pending += __shfl_down(pending, off, 64); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { if (pending) atomicAdd(&ldsCount[0], pending); if (skipped) atomicAdd(&ldsCount[1], skipped); }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (ldsCount[0]) atomicAdd(&stats->rays_shadow, (unsigned long long)ldsCount[0]);
        if (ldsCount[1]) atomicAdd(&stats->rays_shadow_skipped, (unsigned long long)ldsCount[1]);
    }
}

// k_shadow for the scene every BASELINE config has: exactly one light, a sun (culled traversal, no counters).  All
// shadow rays share the direction -light.dir, so its reciprocal and its part of the plain-ray test are computed once per
// thread instead of once per ray (three IEEE divisions), every queued entry awaits its ray (no flag to read), and none
// of the point-light state (shade location, distance to the light, distance limit) is carried: 64 instead of 75
// registers, i.e. 8 instead of 6 waves per SIMD.  Same arithmetic per ray as occluded_fast<.., true> with kind = 1.
__device__ __forceinline__ void shadow_sun_batch(const SceneView& S, const FrameParams& F, const PathBuffers& P, const uint32_t* __restrict__ count,
                                                 int bounce, uint2* lds, uint32_t bid, uint32_t nblk) {
    const Stack stk = make_stack(lds, P);
    const uint32_t n = *count;
    const RefLight Lg = F.lights[0];
    const v3 d = -mk(Lg.dir.x, Lg.dir.y, Lg.dir.z);
    const v3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const float big = 3.0e38f;
    const bool dPlain = fabsf(d.x) < big && fabsf(d.y) < big && fabsf(d.z) < big && d.x != 0.f && d.y != 0.f && d.z != 0.f;
    LogEntry* const logBase = log_plane(P, bounce).base;
    for (uint32_t i = bid * RT_BLOCK + threadIdx.x; i < n; i += nblk * RT_BLOCK) {
        const float4 o4 = P.shadowRec[i];
        const uint32_t slot = __float_as_uint(o4.w);
        const v3 o = mk(o4.x, o4.y, o4.z);
        const bool plain = dPlain && fabsf(o.x) < big && fabsf(o.y) < big && fabsf(o.z) < big;      // == ray_is_plain(o, d)
        uint32_t nNodes = 0, nTris = 0;
        bool occ;
        if (!__all(plain)) {
            occ = occluded_fast<false, false>(S, o, d, 1, mk(0.f, 0.f, 0.f), 0.f, stk, nNodes, nTris);
        } else {
            occ = false;
            float te;
            if (slab<true>(o, inv, S.rootLo, S.rootHi, &te)) {
                int sp = 0, cur = S.rootChild;
                const RayPairs R = make_pairs(o, inv);
                while (cur != RT_DONE) {
                    while (cur >= 0) {
                        const float4* np = reinterpret_cast<const float4*>(S.nodes + cur);
                        float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                        int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
                        float t0, t1, x0, x1;
                        slab_pair(R, n0, n1, n2, t0, x0, t1, x1);
                        const bool p0 = x0 >= t0 && x0 >= 0.0f, p1 = x1 >= t1 && x1 >= 0.0f;
                        if (p0 && p1) {
                            const bool swap = t1 < t0;
                            uint2 e; e.x = (uint32_t)(swap ? c0 : c1); e.y = 0u; stk.push(sp, e);
                            cur = swap ? c1 : c0;
                        }
                        else if (p0) cur = c0;
                        else if (p1) cur = c1;
                        else if (sp > 0) cur = (int)stk.pop(sp).x;
                        else cur = RT_DONE;
                    }
                    if (cur != RT_DONE) {
                        const int code = -cur - 1;
                        const int first = code >> 3, cnt = code & 7;
                        const float4* tp = reinterpret_cast<const float4*>(S.tris + first);
                        for (int k = 0; k < cnt; k++) {
                            HitRec h; h.dist = 3.40282347e+38f; h.idx = -1; h.t = 0.f; h.ref = -1; h.loc = mk(0.f, 0.f, 0.f);
                            test_tri<true>(tp[3 * k], tp[3 * k + 1], tp[3 * k + 2], first + k, o, d, h);
                            if (h.idx >= 0) occ = true;
                        }
                        if (occ) cur = RT_DONE;
                        else if (sp > 0) cur = (int)stk.pop(sp).x;
                        else cur = RT_DONE;
                    }
                }
            }
        }
        if (occ) logBase[slot] = log_entry(0.f, 0.f, 0.f);      // an occluded light adds nothing (PathBuffers::contrib)
    }
}

__global__ __launch_bounds__(RT_BLOCK) void k_shadow_sun(SceneView S, FrameParams F, PathBuffers P, const uint32_t* __restrict__ count,
                                                         int bounce) {
    __shared__ uint2 lds[RT_LDS_STACK * RT_BLOCK];
    shadow_sun_batch(S, F, P, count, bounce, lds, blockIdx.x, gridDim.x);
}

// shader.comp:453-484: the shadow lineCast of every log entry awaiting one; the entry becomes "add" (2) or "none" (0).
template <bool EXACT, bool COUNT>
__device__ __forceinline__ void shadow_batch(const SceneView& S, const FrameParams& F, const PathBuffers& P, const uint32_t* __restrict__ count,
                                             int bounce, DeviceStats* stats, uint2* lds, uint32_t bid, uint32_t nblk) {
    const Stack stk = make_stack(lds, P);
    uint32_t n = *count;
    uint32_t nNodes = 0, nTris = 0;
    for (uint32_t i = bid * RT_BLOCK + threadIdx.x; i < n; i += nblk * RT_BLOCK) {
        const float4 o4 = P.shadowRec[i];
        const uint32_t slot = __float_as_uint(o4.w);      // the path's log index (log_plane): its slot, or its queue position at a late bounce
        v3 o = mk(o4.x, o4.y, o4.z), hitLoc = mk(0.f, 0.f, 0.f);
        if (F.anyPoint) { float4 h4 = P.shadeLoc[slot]; hitLoc = mk(h4.x, h4.y, h4.z); }
        const LogPlane LP = log_plane(P, bounce);
        LogEntry* logp = LP.base + slot;
        for (int l = 0; l < F.numLights; l++) {
            LogEntry* ep = &logp[(size_t)l * LP.stride];
            const RefLight Lg = light_at(S, F, l);
            float4 L0 = make_float4(Lg.loc.x, Lg.loc.y, Lg.loc.z, Lg.strength), L1 = make_float4(Lg.dir.x, Lg.dir.y, Lg.dir.z, 0.f);
            uint32_t ltype = Lg.lightType;
            // Which entries await a shadow ray: those of a light type the shader casts for (shader.comp:462,476) whose
            // contribution is not exactly zero (k_shade logs +0 for N.L <= 0 and for contributions of +-0, and does not cast
            // for them).  One light: a queued slot always awaits its shadow ray, nothing to read.
            if (F.numLights > 1) {
                if (ltype > 1u) continue;
                const LogEntry e = *ep;
                if (e.x == 0.f && e.y == 0.f && e.z == 0.f) continue;
            }
            v3 dirToLight; float distToLight = 0.f;
            if (ltype == 0u) {
                v3 toL = mk(L0.x, L0.y, L0.z) - hitLoc;
                dirToLight = normalize3(toL);
                distToLight = length3(toL);
            } else dirToLight = -mk(L1.x, L1.y, L1.z);
            bool occ;
            const bool plain = __all(ray_is_plain(o, dirToLight));
            if (EXACT) {
                HitRec best;
                bool h = plain ? closest_hit<true, COUNT, true>(S, o, dirToLight, stk, best, nNodes, nTris)
                               : closest_hit<true, COUNT, false>(S, o, dirToLight, stk, best, nNodes, nTris);
                if (ltype == 0u) { v3 dl = best.loc - hitLoc; occ = h && (sqrtf(dot3(dl, dl)) < distToLight); }
                else occ = h;
            } else {
                occ = plain ? occluded_fast<COUNT, true>(S, o, dirToLight, (int)ltype, hitLoc, distToLight, stk, nNodes, nTris)
                            : occluded_fast<COUNT, false>(S, o, dirToLight, (int)ltype, hitLoc, distToLight, stk, nNodes, nTris);
            }
            if (occ) *ep = log_entry(0.f, 0.f, 0.f);
        }
    }
    if (COUNT) {
        __shared__ uint32_t red[2];
        if (threadIdx.x < 2) red[threadIdx.x] = 0;
        __syncthreads();
        for (int off = 32; off > 0; off >>= 1) { nNodes += __shfl_down(nNodes, off, 64); nTris += __shfl_down(nTris, off, 64); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&red[0], nNodes); atomicAdd(&red[1], nTris); }
        __syncthreads();
        if (threadIdx.x == 0 && (red[0] | red[1])) {
            atomicAdd(&stats->shadow_node_visits, (unsigned long long)red[0]);
            atomicAdd(&stats->shadow_tri_tests, (unsigned long long)red[1]);
        }
    }
}

template <bool EXACT, bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) void k_shadow(SceneView S, FrameParams F, PathBuffers P, const uint32_t* __restrict__ count,
                                                     int bounce, DeviceStats* stats) {
    __shared__ uint2 lds[RT_LDS_STACK * RT_BLOCK];
    shadow_batch<EXACT, COUNT>(S, F, P, count, bounce, stats, lds, blockIdx.x, gridDim.x);
}

// shader.comp:332 and 521: `color.rgb += ...` replayed per pixel in the shader's own order — samples ascending,
// bounces ascending, lights ascending — from the pass's contribution log.  One thread per pixel slot.
// The sum itself is a serial chain, but its operands need not arrive one round trip at a time: the depths of RT_RESOLVE_AHEAD
// samples are fetched together, then (one light: the common case) the first two bounces' entries of all of them, which is
// where nearly all entries are; deeper bounces are read as the replay reaches them.
#ifndef RT_RESOLVE_AHEAD
#define RT_RESOLVE_AHEAD 4
#endif
__device__ __forceinline__ void resolve_add(float4& acc, LogEntry e) {
    acc.x = acc.x + e.x; acc.y = acc.y + e.y; acc.z = acc.z + e.z;
}
// bounces [b0, d) of the path in `slot`, in order
__device__ __forceinline__ void resolve_from(const PathBuffers& P, float4& acc, uint32_t slot, uint32_t b0, uint32_t d) {
    const uint32_t dDense = d < P.denseBounces ? d : P.denseBounces;
    for (uint32_t b = b0; b < dDense; b++) {
        const LogEntry* logp = P.contrib + (size_t)b * P.lightSlots * P.nslots + slot;
        for (uint32_t l = 0; l < P.lightSlots; l++) resolve_add(acc, logp[(size_t)l * P.nslots]);
    }
    if (d > dDense) {
        // the path's later bounces: from its queue position at bounce denseBounces along the chain of next positions
        // (a position beyond the log only occurs in a frame that overflowed it, which the host discards)
        uint32_t i = P.lateHead[slot];
        for (uint32_t b = P.denseBounces; b < d && i < P.lateCap; b++) {
            const size_t plane = (size_t)(b - P.denseBounces);
            const uint32_t nxt = P.lateNext[plane * P.lateCap + i];
            if (b >= b0) {
                const LogEntry* logp = P.late + plane * P.lightSlots * P.lateCap + i;
                for (uint32_t l = 0; l < P.lightSlots; l++) resolve_add(acc, logp[(size_t)l * P.lateCap]);
            }
            i = nxt;
        }
    }
}
__global__ __launch_bounds__(RT_BLOCK) void k_resolve(PathBuffers P, int samplesInPass, int maxBounces, int hasSun, float initialStrength) {
    uint32_t ps = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (ps >= P.slotsPerSample) return;
    const LogEntry miss0 = miss_entry(hasSun, mk(initialStrength, initialStrength, initialStrength));      // a camera ray's miss (k_shade, bounce 0)
    float4 acc = P.accum[ps];
    const uint32_t S = (uint32_t)samplesInPass;
    // (two bounces ahead only where both are planes of the dense part)
    const bool ahead = P.lightSlots == 1u && P.denseBounces >= 2u && maxBounces >= 2;
    for (uint32_t sm0 = 0; sm0 < S; sm0 += RT_RESOLVE_AHEAD) {
        uint32_t slot[RT_RESOLVE_AHEAD], d[RT_RESOLVE_AHEAD];
#pragma unroll
        for (int k = 0; k < RT_RESOLVE_AHEAD; k++) {
            slot[k] = slot_of(P, sm0 + (uint32_t)k < S ? sm0 + (uint32_t)k : S - 1u, ps);
            d[k] = sm0 + (uint32_t)k < S ? P.depth[slot[k]] : 0u;
        }
        if (ahead) {
            LogEntry e0[RT_RESOLVE_AHEAD], e1[RT_RESOLVE_AHEAD];
#pragma unroll
            for (int k = 0; k < RT_RESOLVE_AHEAD; k++) {
                e0[k] = log_entry(0.f, 0.f, 0.f); e1[k] = e0[k];
                if (d[k] & RT_DEPTH_MISS0) { e0[k] = miss0; d[k] = 1u; }
                else if (d[k] > 0u) e0[k] = P.contrib[slot[k]];
                if (d[k] > 1u) e1[k] = P.contrib[(size_t)P.nslots + slot[k]];
            }
#pragma unroll
            for (int k = 0; k < RT_RESOLVE_AHEAD; k++) {
                resolve_add(acc, e0[k]);          // (+0 where there is no such bounce: nothing changes)
                resolve_add(acc, e1[k]);
                if (d[k] > 2u) resolve_from(P, acc, slot[k], 2u, d[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < RT_RESOLVE_AHEAD; k++) {
                if (d[k] & RT_DEPTH_MISS0) resolve_add(acc, miss0);
                else resolve_from(P, acc, slot[k], 0u, d[k]);
            }
        }
    }
    P.accum[ps] = acc;
}

// Image defaults the shader stores for pixels whose primary ray misses (shader.comp:299-301,534-536).
__global__ __launch_bounds__(RT_BLOCK) void k_init_frame(FrameParams F, PathBuffers P, float4* imgAlbedo, float4* imgNormal, float4* imgDepth) {
    uint32_t slot = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (slot >= P.slotsPerSample) return;
    P.accum[slot] = make_float4(0.f, 0.f, 0.f, 1.f);
    int px, py;
    slot_to_pixel(F, slot, &px, &py);
    if (px < F.rx + F.rw && py < F.ry + F.rh && imgAlbedo) {
        size_t pix = (size_t)py * F.width + px;
        imgAlbedo[pix] = make_float4(0.f, 0.f, 0.f, 1.f);
        imgNormal[pix] = make_float4(0.5f, 0.5f, 0.5f, 1.f);
        imgDepth[pix] = make_float4(0.f, 0.f, 0.f, 1.f);
    }
}

// shader.comp:533 — un-swizzle the tile-ordered accumulators into the row-major colour image.
// alpha: 1 (the shader's vec4(color, 1)) for the call whose sample range holds sample 0, 0 for the other sample ranges of
// a split frame, so that the element-wise sum that merges partial renders leaves alpha at 1 like the AOV images'.
__global__ __launch_bounds__(RT_BLOCK) void k_finalize(FrameParams F, PathBuffers P, float4* imgColor, float alpha) {
    uint32_t slot = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (slot >= P.slotsPerSample) return;
    int px, py;
    slot_to_pixel(F, slot, &px, &py);
    if (px < F.rx + F.rw && py < F.ry + F.rh) {
        float4 c = P.accum[slot];
        imgColor[(size_t)py * F.width + px] = make_float4(c.x, c.y, c.z, alpha);
    }
}

// rt_trace_rays: caller rays become entries 0..n-1 of the parity-0 ray buffer.
__global__ __launch_bounds__(RT_BLOCK) void k_load_rays(PathBuffers P, const float* origins, const float* dirs, uint32_t n, uint32_t* countOut) {
    uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i < n) {
        P.ray[0][2u * i] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], dirs[3 * i]);
        P.ray[0][2u * i + 1u] = make_float4(dirs[3 * i + 1], dirs[3 * i + 2], 0.f, __uint_as_float(i));
    }
    if (i == 0) *countOut = n;
}

}  // namespace rt

#include "rt_trace_refill.hip.h"
