// gfx950 traversal kernels of traversalMode 0 with the top of the BVH resident in LDS (included by rt_kernels.hip.h).
//
// Measured on MI355X (tools/ubench/gather_rate.hip, and by adding one redundant load / 24 redundant multiplies per
// node visit to k_extend: +20 % vs +3.6 % time): what bounds the batch traversal kernels on a cache-resident scene is
// not the vector ALU but the CU's vector-memory path — a per-lane gather of 64-byte node records from L2 costs
// ~2.9 CU-cycles per lane (every L1 miss fills a whole 128-byte line at 64 B/clk), ~185 cycles per 64-lane node step,
// and node records are ~70 % of a ray's vector-memory instructions.  These kernels therefore keep the first
// RT_LDS_NODES nodes of the compact BVH (breadth-first order: the top of the tree; the whole tree for the bundled
// scenes up to scene 4) in LDS — one copy per CU, shared by a 1024-thread workgroup — as four arrays (three 16-byte
// box words + one 8-byte child pair), so the lanes of one ds_read spread over the banks.  Nodes below the resident part
// and all triangle records still come through the vector cache.
//
// One workgroup per CU means 16 waves per CU, so memory latency has to be cut rather than hidden: the first triangle
// record of a leaf is requested the moment a lane's descent reaches the leaf (the wave keeps stepping the other
// lanes through LDS nodes meanwhile) and record k + 1 is requested before record k is tested.
//
// Per-ray arithmetic: that of closest_hit<false,..,true> / occluded_fast<..,true> (shader.comp:104-125, 171-184,
// 243-290) — same expressions, same order.  Waves holding a ray that could produce a NaN in a slab test take the
// select-form routines above.
#pragma once

namespace rt {

#define RT_LDS_BLOCK 1024
#define RT_LDS_DEPTH 6            // per-lane stack entries in LDS (deeper ones: global overflow area)
#define RT_LDS_NODES 2000         // nodes resident in LDS: 2000 x 56 B + 1024 x 6 x 8 B of stack = 161,152 B of 163,840

struct LdsImage {
    float4 q0[RT_LDS_NODES], q1[RT_LDS_NODES], q2[RT_LDS_NODES];   // CNode words 0..2 (the two child boxes)
    int2 ch[RT_LDS_NODES];                                         // child0, child1
    uint2 stack[RT_LDS_DEPTH * RT_LDS_BLOCK];
    uint32_t red[4];
};

__device__ __forceinline__ uint32_t stage_nodes(LdsImage& sh, const SceneView& S) {
    const uint32_t nRes = S.n_nodes < (uint32_t)RT_LDS_NODES ? S.n_nodes : (uint32_t)RT_LDS_NODES;
    for (uint32_t i = threadIdx.x; i < nRes; i += RT_LDS_BLOCK) {
        const float4* np = reinterpret_cast<const float4*>(S.nodes + i);
        float4 a = np[0], b = np[1], c = np[2], d = np[3];
        sh.q0[i] = a; sh.q1[i] = b; sh.q2[i] = c;
        sh.ch[i] = make_int2(__float_as_int(d.x), __float_as_int(d.y));
    }
    if (threadIdx.x < 4) sh.red[threadIdx.x] = 0;
    __syncthreads();
    return nRes;
}

__device__ __forceinline__ void fetch_node(const LdsImage& sh, const SceneView& S, uint32_t nRes, int cur,
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

// The leaf a lane stands at: triangle k of the run [first, first + cnt); record first + k is in (or on its way into) A, B, C.
struct LeafCursor {
    int first, cnt, k;
    float4 A, B, C;
    __device__ __forceinline__ void request(const SceneView& S) {
        const float4* tp = reinterpret_cast<const float4*>(S.tris + (first + k));
        A = tp[0]; B = tp[1]; C = tp[2];
    }
};

// Closest hit == lineCast (shader.comp:243-290) for one ray per lane; all lanes of the wave hold plain rays.
template <bool COUNT>
__device__ __forceinline__ void closest_hit_lds(const LdsImage& sh, uint32_t nRes, const SceneView& S, bool live, v3 o, v3 d,
                                                const Stack& stk, HitRec& best, uint32_t& nNodes, uint32_t& nTris, uint32_t* wSteps) {
    const float INF = __int_as_float(0x7f800000);
    best.dist = 3.40282347e+38f; best.t = INF; best.idx = -1; best.ref = -1; best.loc = mk(0.f, 0.f, 0.f);
    float lim = INF;
    int sp = 0, cur = RT_DONE;
    LeafCursor lc; lc.first = 0; lc.cnt = 0; lc.k = 0;
    lc.A = lc.B = lc.C = make_float4(0.f, 0.f, 0.f, 0.f);
    const v3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const RayPairs R = make_pairs(o, inv);

    auto popNext = [&]() {
        cur = RT_DONE;
        while (sp > 0) {
            uint2 e = stk.pop(sp);
            if (__uint_as_float(e.y) > lim) continue;
            cur = (int)e.x;
            break;
        }
    };
    // cur is a leaf code or RT_DONE: stand at the leaf's first triangle and request its record (empty leaves are skipped)
    auto enterLeaf = [&]() {
        while (cur != RT_DONE) {
            if (cur >= 0) break;
            int code = -cur - 1;
            lc.first = code >> 3; lc.cnt = code & 7; lc.k = 0;
            if (lc.cnt > 0) { lc.request(S); break; }
            popNext();
        }
    };

    if (live) {
        float te;
        if (COUNT) nNodes++;
        if (slab<true>(o, inv, S.rootLo, S.rootHi, &te)) { cur = S.rootChild; if (cur < 0) enterLeaf(); }
    }
    while (cur != RT_DONE) {
        while (cur >= 0) {
            float4 n0, n1, n2; int c0, c1;
            fetch_node(sh, S, nRes, cur, n0, n1, n2, c0, c1);
            if (COUNT) { nNodes++; if (wSteps && first_active_lane()) wSteps[0]++; }
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
            if (cur < 0) enterLeaf();           // leaf reached (or done): its first record is requested now
        }
        if (cur != RT_DONE) {                   // at a leaf with cnt > 0
            while (lc.k < lc.cnt) {
                if (COUNT) { nTris++; if (wSteps && first_active_lane()) wSteps[1]++; }
                const float4 A = lc.A, B = lc.B, C = lc.C;
                const int idx = lc.first + lc.k;
                lc.k++;
                if (lc.k < lc.cnt) lc.request(S);
                float t;
                if (tri_t(A, B, C, o, d, &t)) {
                    v3 h = o + d * t;
                    v3 oh = o - h;
                    float newDist = dot3(oh, oh);
                    if (newDist < best.dist || (newDist == best.dist && idx < best.idx)) {
                        best.dist = newDist; best.t = t; best.idx = idx; best.ref = __float_as_int(C.y); best.loc = h;
                        lim = t + t * 1e-4f + S.cullAbs;
                    }
                }
            }
            popNext();
            if (cur < 0) enterLeaf();
        }
    }
}

template <bool COUNT, bool PRIMARY>
__global__ __launch_bounds__(RT_LDS_BLOCK) void k_extend_lds(SceneView S, FrameParams F, PathBuffers P,
                                                             const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count,
                                                             int rBase, uint32_t rayCount, DeviceStats* stats) {
    __shared__ LdsImage sh;
    const uint32_t nRes = stage_nodes(sh, S);
    const Stack stk = make_stack(sh.stack, P, RT_LDS_DEPTH, RT_LDS_BLOCK);
    const uint32_t n = *count;
    const uint32_t nUp = (n + 63u) & ~63u;        // whole waves: the traversal routine is entered wave-uniformly
    uint32_t nNodes = 0, nTris = 0, maxSteps = 0;
    uint32_t wSteps[2] = {0u, 0u};
    for (uint32_t i = blockIdx.x * RT_LDS_BLOCK + threadIdx.x; i < nUp; i += gridDim.x * RT_LDS_BLOCK) {
        uint32_t slot = 0;
        v3 ro = mk(0.f, 0.f, 0.f), rd = mk(1.f, 1.f, 1.f);
        bool live = i < n;
        if (live) {
            if (PRIMARY) {
                slot = i;
                int px, py, r;
                ro = ld3(F.camPos);
                if (!primary_ray(F, P, rBase, slot, &px, &py, &r, &rd)) {
                    P.hit[slot] = make_float4(0.f, 0.f, 0.f, __int_as_float(-2));
                    live = false; rd = mk(1.f, 1.f, 1.f);
                }
            } else {
                slot = queue[i];
                float4 o4 = P.rayO[slot], d4 = P.rayD[slot];
                ro = mk(o4.x, o4.y, o4.z); rd = mk(d4.x, d4.y, d4.z);
            }
        }
        HitRec best;
        uint32_t before = nNodes + nTris;
        if (__all(!live || ray_is_plain(ro, rd))) closest_hit_lds<COUNT>(sh, nRes, S, live, ro, rd, stk, best, nNodes, nTris, wSteps);
        else if (live) closest_hit<false, COUNT, false>(S, ro, rd, stk, best, nNodes, nTris, wSteps);
        if (COUNT) { uint32_t steps = nNodes + nTris - before; if (steps > maxSteps) maxSteps = steps; }
        if (live) P.hit[slot] = make_float4(best.loc.x, best.loc.y, best.loc.z, __int_as_float(best.ref));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        atomicAdd(PRIMARY ? &stats->rays_primary : &stats->rays_bounce, (unsigned long long)(rayCount ? rayCount : n));
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {
            nNodes += __shfl_down(nNodes, off, 64); nTris += __shfl_down(nTris, off, 64);
            wSteps[0] += __shfl_down(wSteps[0], off, 64); wSteps[1] += __shfl_down(wSteps[1], off, 64);
        }
        for (int off = 32; off > 0; off >>= 1) { uint32_t o2 = __shfl_down(maxSteps, off, 64); if (o2 > maxSteps) maxSteps = o2; }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&sh.red[0], nNodes); atomicAdd(&sh.red[1], nTris); atomicAdd(&sh.red[2], wSteps[0]); atomicAdd(&sh.red[3], wSteps[1]);
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

// Any-hit for one shadow ray per lane (see occluded_fast); all lanes of the wave hold plain rays.
template <bool COUNT>
__device__ __forceinline__ bool occluded_lds(const LdsImage& sh, uint32_t nRes, const SceneView& S, bool live, v3 o, v3 d, int kind,
                                             v3 shadeLoc, float distToLight, const Stack& stk, uint32_t& nNodes, uint32_t& nTris) {
    const float INF = __int_as_float(0x7f800000);
    const float lim = (kind == 0) ? (distToLight + distToLight * 1e-3f + S.cullAbs * 4.0f) : INF;
    int sp = 0, cur = RT_DONE;
    bool occ = false;
    LeafCursor lc; lc.first = 0; lc.cnt = 0; lc.k = 0;
    lc.A = lc.B = lc.C = make_float4(0.f, 0.f, 0.f, 0.f);
    const v3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const RayPairs R = make_pairs(o, inv);
    auto popNext = [&]() { cur = (sp > 0) ? (int)stk.pop(sp).x : RT_DONE; };
    auto enterLeaf = [&]() {
        while (cur != RT_DONE) {
            if (cur >= 0) break;
            int code = -cur - 1;
            lc.first = code >> 3; lc.cnt = code & 7; lc.k = 0;
            if (lc.cnt > 0) { lc.request(S); break; }
            popNext();
        }
    };
    if (live) {
        float te;
        if (COUNT) nNodes++;
        if (slab<true>(o, inv, S.rootLo, S.rootHi, &te)) { cur = S.rootChild; if (cur < 0) enterLeaf(); }
    }
    while (cur != RT_DONE) {
        while (cur >= 0) {
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
            if (cur < 0) enterLeaf();
        }
        if (cur != RT_DONE) {
            while (lc.k < lc.cnt) {
                if (COUNT) nTris++;
                const float4 A = lc.A, B = lc.B, C = lc.C;
                lc.k++;
                if (lc.k < lc.cnt) lc.request(S);
                float t;
                if (tri_t(A, B, C, o, d, &t)) {
                    if (kind == 1) occ = true;
                    else { v3 h = o + d * t; v3 dl = h - shadeLoc; if (sqrtf(dot3(dl, dl)) < distToLight) occ = true; }
                }
            }
            if (occ) cur = RT_DONE;
            else { popNext(); if (cur < 0) enterLeaf(); }
        }
    }
    return occ;
}

// shader.comp:453-484: the shadow lineCast of every log entry awaiting one; the entry becomes "add" (2) or "none" (0).
template <bool COUNT>
__global__ __launch_bounds__(RT_LDS_BLOCK) void k_shadow_lds(SceneView S, FrameParams F, PathBuffers P, const uint32_t* __restrict__ count,
                                                             int bounce, DeviceStats* stats) {
    __shared__ LdsImage sh;
    const uint32_t nRes = stage_nodes(sh, S);
    const Stack stk = make_stack(sh.stack, P, RT_LDS_DEPTH, RT_LDS_BLOCK);
    const uint32_t n = *count;
    const uint32_t nUp = (n + 63u) & ~63u;
    uint32_t nNodes = 0, nTris = 0;
    for (uint32_t i = blockIdx.x * RT_LDS_BLOCK + threadIdx.x; i < nUp; i += gridDim.x * RT_LDS_BLOCK) {
        const bool have = i < n;
        uint32_t slot = 0;
        v3 o = mk(0.f, 0.f, 0.f), hitLoc = mk(0.f, 0.f, 0.f);
        if (have) {
            slot = P.shadowQueue[i];
            float4 o4 = P.rayO[slot], h4 = P.hit[slot];
            o = mk(o4.x, o4.y, o4.z); hitLoc = mk(h4.x, h4.y, h4.z);
        }
        float4* logp = P.contrib + (size_t)bounce * P.lightSlots * P.nslots + slot;
        for (int l = 0; l < F.numLights; l++) {
            float* wp = &logp[(size_t)l * P.nslots].w;
            const bool live = have && *wp == 1.f;
            const float4* lp = reinterpret_cast<const float4*>(S.lights + l);
            float4 L0 = lp[0], L1 = lp[1];
            const uint32_t ltype = __float_as_uint(L1.w);
            v3 dirToLight = mk(1.f, 1.f, 1.f); float distToLight = 0.f;
            if (live) {
                if (ltype == 0u) {
                    v3 toL = mk(L0.x, L0.y, L0.z) - hitLoc;
                    dirToLight = normalize3(toL);
                    distToLight = length3(toL);
                } else dirToLight = -mk(L1.x, L1.y, L1.z);
            }
            if (!__any(live)) continue;
            bool occ = false;
            if (__all(!live || ray_is_plain(o, dirToLight)))
                occ = occluded_lds<COUNT>(sh, nRes, S, live, o, dirToLight, (int)ltype, hitLoc, distToLight, stk, nNodes, nTris);
            else if (live)
                occ = occluded_fast<COUNT, false>(S, o, dirToLight, (int)ltype, hitLoc, distToLight, stk, nNodes, nTris);
            if (live) *wp = occ ? 0.f : 2.f;
        }
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { nNodes += __shfl_down(nNodes, off, 64); nTris += __shfl_down(nTris, off, 64); }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { atomicAdd(&sh.red[0], nNodes); atomicAdd(&sh.red[1], nTris); }
        __syncthreads();
        if (threadIdx.x == 0 && (sh.red[0] | sh.red[1])) {
            atomicAdd(&stats->shadow_node_visits, (unsigned long long)sh.red[0]);
            atomicAdd(&stats->shadow_tri_tests, (unsigned long long)sh.red[1]);
        }
    }
}

}  // namespace rt
