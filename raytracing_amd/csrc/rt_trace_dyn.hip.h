// gfx950 traversal kernels of traversalMode 0 (included by rt_kernels.hip.h): BVH top resident in LDS, persistent
// waves with dynamic ray fetch.
//
// What bounds traversal of a cache-resident scene on MI355X is the CU's vector-memory path: every node visit is a
// per-lane gather of one 64-byte record (4 x global_load_dwordx4 per lane, one L1 tag look-up per lane and
// instruction), ~80 % of all vector-memory work of a ray.  These kernels therefore keep the first RT_DYN_NODES nodes
// of the compact BVH (breadth-first order, so that is the top of the tree; all of it for the bundled scenes up to
// scene 4) in LDS, one copy per CU shared by a 1024-thread workgroup, as four separate arrays (three 16-byte box
// words + one 8-byte child pair) so that the lanes of a ds_read spread over the banks.  Node visits below the resident
// part, and all triangle records, still come through the vector cache.
//
// With node fetches out of the way the vector ALU is next, and incoherent rays leave most lanes of a fixed 64-ray
// batch idle (39 % active in the batch kernels).  So a wave is persistent: whenever `refillIdle` or more of its lanes
// have finished their ray it hands those lanes new rays and goes on.  Work distribution uses no global atomics (one
// counter word costs ~30 ns per grab on MI355X, measured): workgroup b owns the 1024-ray super-groups b, b + grid,
// b + 2 grid ... of the queue (interleaved, so the resident waves work on one compact window of the queue and of the
// path-state arrays) and its 16 waves take 64-ray groups from them through one LDS counter.
//
// The per-ray arithmetic is that of closest_hit<false,..,true> / occluded_fast<..,true> (shader.comp:104-125,
// 171-184, 243-290): same expressions in the same order; the two child boxes of a node are evaluated with packed FP32
// subtract / multiply (v_pk_add_f32, v_pk_mul_f32), which round each half exactly like the scalar forms.  Rays that
// could produce a NaN in a slab test (zero / non-finite direction component) take the select-form routines, one ray
// at a time.
#pragma once

namespace rt {

#define RT_DYN_BLOCK 1024
#define RT_DYN_STACK 6            // per-lane stack entries in LDS (deeper ones: global overflow area)
#define RT_DYN_NODES 2000         // BVH nodes resident in LDS: 2000 x 56 B + 1024 x 6 x 8 B stack = 161,152 B of 163,840

typedef float f2 __attribute__((ext_vector_type(2)));

// Ray origin and reciprocal direction as the register pairs the packed slab test consumes.
struct RayPairs { f2 oxy, ozx, oyz, ixy, izx, iyz; };

__device__ __forceinline__ RayPairs make_pairs(v3 o, v3 inv) {
    RayPairs R;
    R.oxy = f2{o.x, o.y}; R.ozx = f2{o.z, o.x}; R.oyz = f2{o.y, o.z};
    R.ixy = f2{inv.x, inv.y}; R.izx = f2{inv.z, inv.x}; R.iyz = f2{inv.y, inv.z};
    return R;
}

// slab<true> on both child boxes of a CNode (n0..n2 = its first three 16-byte words).
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

// shader.comp:104-125 on a pre-transformed triangle: true + t on a hit.
__device__ __forceinline__ bool tri_t(float4 A, float4 B, float4 C, v3 o, v3 d, float* tOut) {
    v3 v0 = mk(A.x, A.y, A.z), e1 = mk(A.w, B.x, B.y), e2 = mk(B.z, B.w, C.x);
    v3 pv = cross3(d, e2);
    float det = dot3(e1, pv);
    if (fabsf(det) < 1e-8f) return false;
    float invDet = 1.0f / det;
    v3 s = o - v0;
    float u = dot3(s, pv) * invDet;
    if (u < 0.0f || u > 1.0f) return false;
    v3 q = cross3(s, e1);
    float v = dot3(d, q) * invDet;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = dot3(e2, q) * invDet;
    if (t < 1e-8f) return false;
    *tOut = t;
    return true;
}

// LDS image of one workgroup: resident BVH top, traversal stacks, work cursor.
struct DynShared {
    float4 q0[RT_DYN_NODES], q1[RT_DYN_NODES], q2[RT_DYN_NODES];   // CNode words 0..2 (the two child boxes)
    int2 ch[RT_DYN_NODES];                                         // child0, child1
    uint2 stack[RT_DYN_STACK * RT_DYN_BLOCK];
    uint32_t cursor;                                               // 64-item groups handed out so far
    uint32_t red[4];
};

__device__ __forceinline__ uint32_t stage_nodes(DynShared& sh, const SceneView& S) {
    const uint32_t nRes = S.n_nodes < (uint32_t)RT_DYN_NODES ? S.n_nodes : (uint32_t)RT_DYN_NODES;
    for (uint32_t i = threadIdx.x; i < nRes; i += RT_DYN_BLOCK) {
        const float4* np = reinterpret_cast<const float4*>(S.nodes + i);
        float4 a = np[0], b = np[1], c = np[2], d = np[3];
        sh.q0[i] = a; sh.q1[i] = b; sh.q2[i] = c;
        sh.ch[i] = make_int2(__float_as_int(d.x), __float_as_int(d.y));
    }
    if (threadIdx.x == 0) { sh.cursor = 0; sh.red[0] = 0; sh.red[1] = 0; sh.red[2] = 0; sh.red[3] = 0; }
    __syncthreads();
    return nRes;
}

__device__ __forceinline__ void fetch_node(const DynShared& sh, const SceneView& S, uint32_t nRes, int cur,
                                           float4& n0, float4& n1, float4& n2, int& c0, int& c1) {
    if ((uint32_t)cur < nRes) {
        n0 = sh.q0[cur]; n1 = sh.q1[cur]; n2 = sh.q2[cur];
        int2 c = sh.ch[cur];
        c0 = c.x; c1 = c.y;
    } else {
        const float4* np = reinterpret_cast<const float4*>(S.nodes + cur);
        n0 = np[0]; n1 = np[1]; n2 = np[2];
        float4 n3 = np[3];
        c0 = __float_as_int(n3.x); c1 = __float_as_int(n3.y);
    }
}

// A wave's source of work items (see the header comment).  All members are wave-uniform.
struct WavePool {
    uint32_t next, end;      // unassigned items of the current 64-item group
    uint32_t n;
    __device__ __forceinline__ bool empty() const { return next >= end; }
    __device__ __forceinline__ void open_group(DynShared& sh) {
        uint32_t c = 0;
        if ((threadIdx.x & 63u) == 0u) c = atomicAdd(&sh.cursor, 1u);
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        // group c of this workgroup = sub-group (c % 16) of its super-group (c / 16)
        unsigned long long g = ((unsigned long long)(c >> 4) * gridDim.x + blockIdx.x) * 16ull + (c & 15u);
        unsigned long long b = g * 64ull;
        next = b < n ? (uint32_t)b : n;
        end = (n - next < 64u) ? n : next + 64u;
    }
    // after handing out k items: move on to the next group when this one is used up
    __device__ __forceinline__ void take(DynShared& sh, uint32_t k) {
        next += k;
        if (next >= end && next < n) open_group(sh);
    }
};

__device__ __forceinline__ uint32_t lane_prefix(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// ---------------------------------------------------------------------------------------------
// Vote-scheduled traversal.  A lane is in one of three states: at an internal node (cur >= 0), at triangle k of a
// leaf (cur < 0, cur != RT_DONE; the record is already on its way into A, B, C), or idle (RT_DONE, !active).  Every
// scheduler iteration the wave executes ONE body — the node step or the triangle step — for the lanes in that
// state, choosing the state more lanes are in; the others wait one turn.  Lanes of a wave therefore need not agree on
// the sequence of node and triangle steps of their rays (in the while-while form every lane waits for the longest
// descent and the fullest leaf: 31 % of the lanes active per instruction on bounce rays), and the triangle records
// requested when a lane enters a leaf arrive while the wave keeps stepping through LDS-resident nodes.
struct LeafCursor {
    int first, cnt, k;        // triangle k of the run [first, first + cnt)
    float4 A, B, C;           // record first + k (requested, perhaps still in flight)
};

__device__ __forceinline__ void request_tri(const SceneView& S, LeafCursor& lc) {
    const float4* tp = reinterpret_cast<const float4*>(S.tris + (lc.first + lc.k));
    lc.A = tp[0]; lc.B = tp[1]; lc.C = tp[2];
}

// Closest hit == lineCast (shader.comp:243-290) for the rays of `queue` (PRIMARY: the camera rays of all slots).
template <bool COUNT, bool PRIMARY>
__global__ __launch_bounds__(RT_DYN_BLOCK) void k_extend_dyn(SceneView S, FrameParams F, PathBuffers P,
                                                             const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count,
                                                             int rBase, uint32_t rayCount, int refillIdle, DeviceStats* stats) {
    __shared__ DynShared sh;
    const uint32_t nRes = stage_nodes(sh, S);
    const Stack stk = make_stack(sh.stack, P, RT_DYN_STACK, RT_DYN_BLOCK);
    const float INF = __int_as_float(0x7f800000);
    WavePool wp; wp.n = *count;
    wp.open_group(sh);
    uint32_t nNodes = 0, nTris = 0, maxSteps = 0, steps0 = 0, wNode = 0, wTri = 0;

    bool active = false;
    uint32_t slot = 0;
    v3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 0.f);
    RayPairs R = make_pairs(o, o);
    float bestT = INF, bestDist = 3.40282347e+38f, lim = INF;
    int bestIdx = -1, bestRef = -1;
    int sp = 0, cur = RT_DONE;
    LeafCursor lc; lc.first = 0; lc.cnt = 0; lc.k = 0;
    lc.A = lc.B = lc.C = make_float4(0.f, 0.f, 0.f, 0.f);

    // cur := next stack entry that survives the distance cull (RT_DONE when the stack is empty); entering a leaf
    // requests its first triangle record
    auto popNext = [&]() {
        cur = RT_DONE;
        while (sp > 0) {
            uint2 e = stk.pop(sp);
            if (__uint_as_float(e.y) > lim) continue;
            cur = (int)e.x;
            break;
        }
    };
    auto enterLeaf = [&]() {      // cur is a leaf code or RT_DONE
        while (cur != RT_DONE && cur < 0) {
            int code = -cur - 1;
            lc.first = code >> 3; lc.cnt = code & 7; lc.k = 0;
            if (lc.cnt > 0) { request_tri(S, lc); break; }
            popNext();            // empty leaf
        }
    };
    auto finish = [&]() {         // cur == RT_DONE: the ray is traced
        v3 h = mk(0.f, 0.f, 0.f);
        if (bestIdx >= 0) h = o + d * bestT;
        P.hit[slot] = make_float4(h.x, h.y, h.z, __int_as_float(bestRef));
        if (COUNT) { uint32_t st = nNodes + nTris - steps0 + 1; if (st > maxSteps) maxSteps = st; }
        active = false;
    };

    for (;;) {
        const unsigned long long idleMask = __ballot(!active);
        const int nIdle = __popcll(idleMask);
        if (!wp.empty() && nIdle >= refillIdle) {
            const uint32_t pre = lane_prefix(idleMask);
            const uint32_t avail = wp.end - wp.next;
            if (!active && pre < avail) {
                const uint32_t i = wp.next + pre;
                bool ok = true;
                if (PRIMARY) {
                    slot = i;
                    int px, py, r;
                    o = ld3(F.camPos);
                    ok = primary_ray(F, P, rBase, slot, &px, &py, &r, &d);
                    if (!ok) P.hit[slot] = make_float4(0.f, 0.f, 0.f, __int_as_float(-2));
                } else {
                    slot = queue[i];
                    float4 o4 = P.rayO[slot], d4 = P.rayD[slot];
                    o = mk(o4.x, o4.y, o4.z); d = mk(d4.x, d4.y, d4.z);
                }
                if (ok) {
                    if (!ray_is_plain(o, d)) {
                        HitRec best;
                        closest_hit<false, COUNT, false>(S, o, d, stk, best, nNodes, nTris);
                        P.hit[slot] = make_float4(best.loc.x, best.loc.y, best.loc.z, __int_as_float(best.ref));
                    } else {
                        v3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        float te;
                        if (COUNT) { nNodes++; steps0 = nNodes + nTris; }
                        if (!slab<true>(o, inv, S.rootLo, S.rootHi, &te)) {
                            P.hit[slot] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
                        } else {
                            R = make_pairs(o, inv);
                            bestT = INF; bestDist = 3.40282347e+38f; lim = INF; bestIdx = -1; bestRef = -1;
                            sp = 0; cur = S.rootChild;
                            active = true;
                            enterLeaf();          // the root may itself be a leaf
                            if (cur == RT_DONE) finish();
                        }
                    }
                }
            }
            wp.take(sh, ((uint32_t)nIdle < avail) ? (uint32_t)nIdle : avail);
            continue;
        }
        if (nIdle == 64) break;
        const bool canRefill = !wp.empty();

        for (;;) {
            const unsigned long long mNode = __ballot(cur >= 0);
            const unsigned long long mTri = __ballot(active && cur < 0);
            if ((mNode | mTri) == 0ull) break;
            if (__popcll(mNode) >= __popcll(mTri)) {
                if (cur >= 0) {
                    float4 n0, n1, n2; int c0, c1;
                    fetch_node(sh, S, nRes, cur, n0, n1, n2, c0, c1);
                    if (COUNT) { nNodes++; if (first_active_lane()) wNode++; }
                    float te0, tx0, te1, tx1;
                    slab_pair(R, n0, n1, n2, te0, tx0, te1, tx1);
                    bool p0 = tx0 >= te0 && tx0 >= 0.0f && !(te0 > lim);
                    bool p1 = tx1 >= te1 && tx1 >= 0.0f && !(te1 > lim);
                    if (p0 && p1) {
                        bool swap = te1 < te0;
                        uint2 e; e.x = (uint32_t)(swap ? c0 : c1); e.y = __float_as_uint(swap ? te0 : te1);
                        stk.push(sp, e);
                        cur = swap ? c1 : c0;
                    } else if (p0) cur = c0;
                    else if (p1) cur = c1;
                    else popNext();
                    if (cur < 0) { enterLeaf(); if (cur == RT_DONE) finish(); }
                }
            } else {
                if (active && cur < 0) {
                    if (COUNT) { nTris++; if (first_active_lane()) wTri++; }
                    float t;
                    const float4 A = lc.A, B = lc.B, C = lc.C;
                    const int idx = lc.first + lc.k;
                    lc.k++;
                    if (lc.k < lc.cnt) request_tri(S, lc);      // next record on its way before this one is tested
                    if (tri_t(A, B, C, o, d, &t)) {
                        v3 h = o + d * t;
                        v3 oh = o - h;
                        float newDist = dot3(oh, oh);
                        if (newDist < bestDist || (newDist == bestDist && idx < bestIdx)) {
                            bestDist = newDist; bestT = t; bestIdx = idx; bestRef = __float_as_int(C.y);
                            lim = bestT + bestT * 1e-4f + S.cullAbs;
                        }
                    }
                    if (lc.k >= lc.cnt) {
                        popNext();
                        if (cur < 0) { enterLeaf(); if (cur == RT_DONE) finish(); }
                    }
                }
            }
            if (canRefill && __popcll(__ballot(!active)) >= refillIdle) break;
        }
    }

    if (blockIdx.x == 0 && threadIdx.x == 0)
        atomicAdd(PRIMARY ? &stats->rays_primary : &stats->rays_bounce, (unsigned long long)(rayCount ? rayCount : wp.n));
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {
            nNodes += __shfl_down(nNodes, off, 64); nTris += __shfl_down(nTris, off, 64);
            wNode += __shfl_down(wNode, off, 64); wTri += __shfl_down(wTri, off, 64);
        }
        for (int off = 32; off > 0; off >>= 1) { uint32_t o2 = __shfl_down(maxSteps, off, 64); if (o2 > maxSteps) maxSteps = o2; }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&sh.red[0], nNodes); atomicAdd(&sh.red[1], nTris); atomicAdd(&sh.red[2], wNode); atomicAdd(&sh.red[3], wTri);
            if (maxSteps) atomicMax(&stats->max_ray_steps, (unsigned long long)maxSteps);
        }
        __syncthreads();
        if (threadIdx.x == 0 && (sh.red[0] | sh.red[1])) {
            atomicAdd(&stats->node_visits, (unsigned long long)sh.red[0]);
            atomicAdd(&stats->tri_tests, (unsigned long long)sh.red[1]);
            if (!PRIMARY) {
                atomicAdd(&stats->bounce_node_visits, (unsigned long long)sh.red[0]);
                atomicAdd(&stats->bounce_tri_tests, (unsigned long long)sh.red[1]);
                atomicAdd(&stats->wave_node_steps, (unsigned long long)sh.red[2]);
                atomicAdd(&stats->wave_tri_steps, (unsigned long long)sh.red[3]);
            }
        }
    }
}

// shader.comp:453-484: the shadow lineCast of every log entry awaiting one; the entry becomes "add" (2) or "none" (0).
// Work item i = (shadow-queue entry i / numLights, light i % numLights).
template <bool COUNT>
__global__ __launch_bounds__(RT_DYN_BLOCK) void k_shadow_dyn(SceneView S, FrameParams F, PathBuffers P, const uint32_t* __restrict__ count,
                                                             int bounce, int refillIdle, DeviceStats* stats) {
    __shared__ DynShared sh;
    const uint32_t nRes = stage_nodes(sh, S);
    const Stack stk = make_stack(sh.stack, P, RT_DYN_STACK, RT_DYN_BLOCK);
    const uint32_t L = (uint32_t)F.numLights;
    const float INF = __int_as_float(0x7f800000);
    WavePool wp; wp.n = *count * L;
    wp.open_group(sh);
    uint32_t nNodes = 0, nTris = 0;

    bool active = false;
    float* wptr = nullptr;
    v3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 0.f), shadeLoc = mk(0.f, 0.f, 0.f);
    RayPairs R = make_pairs(o, o);
    float distToLight = 0.f, lim = INF;
    int kind = 1;
    int sp = 0, cur = RT_DONE;
    LeafCursor lc; lc.first = 0; lc.cnt = 0; lc.k = 0;
    lc.A = lc.B = lc.C = make_float4(0.f, 0.f, 0.f, 0.f);

    auto popNext = [&]() { cur = (sp > 0) ? (int)stk.pop(sp).x : RT_DONE; };
    auto enterLeaf = [&]() {
        while (cur != RT_DONE && cur < 0) {
            int code = -cur - 1;
            lc.first = code >> 3; lc.cnt = code & 7; lc.k = 0;
            if (lc.cnt > 0) { request_tri(S, lc); break; }
            popNext();
        }
    };
    auto finish = [&](bool occ) { *wptr = occ ? 0.f : 2.f; active = false; cur = RT_DONE; };

    for (;;) {
        const unsigned long long idleMask = __ballot(!active);
        const int nIdle = __popcll(idleMask);
        if (!wp.empty() && nIdle >= refillIdle) {
            const uint32_t pre = lane_prefix(idleMask);
            const uint32_t avail = wp.end - wp.next;
            if (!active && pre < avail) {
                const uint32_t i = wp.next + pre;
                const uint32_t q = (L == 1u) ? i : i / L, l = (L == 1u) ? 0u : i % L;
                const uint32_t slot = P.shadowQueue[q];
                float* wp_ = &P.contrib[((size_t)bounce * P.lightSlots + l) * P.nslots + slot].w;
                if (*wp_ == 1.f) {
                    float4 o4 = P.rayO[slot], h4 = P.hit[slot];
                    o = mk(o4.x, o4.y, o4.z); shadeLoc = mk(h4.x, h4.y, h4.z);
                    const float4* lp = reinterpret_cast<const float4*>(S.lights + l);
                    float4 L0 = lp[0], L1 = lp[1];
                    kind = (int)__float_as_uint(L1.w);
                    distToLight = 0.f;
                    if (kind == 0) {
                        v3 toL = mk(L0.x, L0.y, L0.z) - shadeLoc;
                        d = normalize3(toL);
                        distToLight = length3(toL);
                    } else d = -mk(L1.x, L1.y, L1.z);
                    wptr = wp_;
                    if (!ray_is_plain(o, d)) {
                        bool oc = occluded_fast<COUNT, false>(S, o, d, kind, shadeLoc, distToLight, stk, nNodes, nTris);
                        *wptr = oc ? 0.f : 2.f;
                    } else {
                        v3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        float te;
                        if (COUNT) nNodes++;
                        if (!slab<true>(o, inv, S.rootLo, S.rootHi, &te)) {
                            *wptr = 2.f;
                        } else {
                            R = make_pairs(o, inv);
                            lim = (kind == 0) ? (distToLight + distToLight * 1e-3f + S.cullAbs * 4.0f) : INF;
                            sp = 0; cur = S.rootChild;
                            active = true;
                            enterLeaf();
                            if (cur == RT_DONE) finish(false);
                        }
                    }
                }
            }
            wp.take(sh, ((uint32_t)nIdle < avail) ? (uint32_t)nIdle : avail);
            continue;
        }
        if (nIdle == 64) break;
        const bool canRefill = !wp.empty();

        for (;;) {
            const unsigned long long mNode = __ballot(cur >= 0);
            const unsigned long long mTri = __ballot(active && cur < 0);
            if ((mNode | mTri) == 0ull) break;
            if (__popcll(mNode) >= __popcll(mTri)) {
                if (cur >= 0) {
                    float4 n0, n1, n2; int c0, c1;
                    fetch_node(sh, S, nRes, cur, n0, n1, n2, c0, c1);
                    if (COUNT) nNodes++;
                    float te0, tx0, te1, tx1;
                    slab_pair(R, n0, n1, n2, te0, tx0, te1, tx1);
                    bool p0 = tx0 >= te0 && tx0 >= 0.0f && !(te0 > lim);
                    bool p1 = tx1 >= te1 && tx1 >= 0.0f && !(te1 > lim);
                    if (p0 && p1) {      // nearer box first: an occluder, if any, tends to be found sooner
                        bool swap = te1 < te0;
                        uint2 e; e.x = (uint32_t)(swap ? c0 : c1); e.y = 0u; stk.push(sp, e);
                        cur = swap ? c1 : c0;
                    } else if (p0) cur = c0;
                    else if (p1) cur = c1;
                    else popNext();
                    if (cur < 0) { enterLeaf(); if (cur == RT_DONE) finish(false); }
                }
            } else {
                if (active && cur < 0) {
                    if (COUNT) nTris++;
                    float t;
                    const float4 A = lc.A, B = lc.B, C = lc.C;
                    lc.k++;
                    if (lc.k < lc.cnt) request_tri(S, lc);
                    bool occ = false;
                    if (tri_t(A, B, C, o, d, &t)) {
                        if (kind == 1) occ = true;
                        else { v3 h = o + d * t; v3 dl = h - shadeLoc; if (sqrtf(dot3(dl, dl)) < distToLight) occ = true; }
                    }
                    if (occ) finish(true);
                    else if (lc.k >= lc.cnt) {
                        popNext();
                        if (cur < 0) { enterLeaf(); if (cur == RT_DONE) finish(false); }
                    }
                }
            }
            if (canRefill && __popcll(__ballot(!active)) >= refillIdle) break;
        }
    }

    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { nNodes += __shfl_down(nNodes, off, 64); nTris += __shfl_down(nTris, off, 64); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&sh.red[0], nNodes); atomicAdd(&sh.red[1], nTris); }
        __syncthreads();
        if (threadIdx.x == 0 && (sh.red[0] | sh.red[1])) {
            atomicAdd(&stats->shadow_node_visits, (unsigned long long)sh.red[0]);
            atomicAdd(&stats->shadow_tri_tests, (unsigned long long)sh.red[1]);
        }
    }
}

}  // namespace rt
