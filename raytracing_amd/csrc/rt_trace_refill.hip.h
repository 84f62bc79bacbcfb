// gfx950 traversal kernels with dynamic ray fetch (traversalMode 0, knob `refill`; included by rt_kernels.hip.h).
//
// A wave of the batch kernels traces one fixed set of 64 rays and runs until the longest of them is done.  On a scene
// that does not fit the caches (C4: 20 M triangles, 25 nodes per ray, dependent misses to the Infinity Cache / HBM)
// only 13 % of the lanes are active per node step, so a wave has few misses in flight and the kernel is bound by memory
// latency.  Here a wave is persistent instead: whenever `refillIdle` or more of its lanes have finished their ray it
// hands those lanes new rays and goes on, which keeps more independent misses in flight per wave.  Work distribution
// uses no atomics (one contended counter word costs ~30 ns per grab on MI355X): wave w owns the 64-ray groups
// w, w + nWaves, w + 2 nWaves ... of the queue, so the resident waves work on one compact window of the queue and of the
// path-state arrays.  With the leaf parking below the closest-hit kernel is the faster one on the cache-resident scenes too
// (bounce rays; camera rays stay on the batch kernel), and which pair of kernels a scene gets is measured (rt_core.hip).
//
// Per-ray arithmetic: that of closest_hit<false,..,true> / occluded_fast<..,true> (shader.comp:104-125, 171-184,
// 243-290) — same expressions, same order.  Rays that could produce a NaN in a slab test take the select-form routines
// above, one ray at a time.
#pragma once

namespace rt {
__device__ __forceinline__ void shadow_sun_packet(const SceneView& S, const FrameParams& F, const PathBuffers& P, const uint32_t* __restrict__ count,
                                                  int bounce, uint2* lds, uint32_t bid, uint32_t nblk);      // rt_trace_packet.hip.h

// shader.comp:104-125 on a pre-transformed triangle: true + t on a hit.
__device__ __forceinline__ bool tri_t(float4 A, float4 B, float4 C, v3 o, v3 d, float* tOut) {
    asm volatile("" : "+v"(A.x), "+v"(A.y), "+v"(A.z));     // see test_tri: keep the load of v0 in front of the determinant test
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

// This wave's share of the work items [0, n): the 64-item groups wave, wave + nWaves, wave + 2 nWaves ... (grid
// stride), so the waves resident at one time work on one compact window of the queue and of the path-state arrays
// (contiguous per-wave ranges spread ~5000 concurrent streams over HBM pages and were measured 30 % slower).
// No atomics: a work counter shared by all waves costs ~30 ns per grab on MI355X (one L2 atomic unit serves the
// word), which at 64..1024 items per grab was measured to cost more than the lanes it keeps busy.
struct WavePool {
    uint32_t next, end;      // unassigned items of the current group
    uint32_t group, stride, n;
    __device__ __forceinline__ void open_group() {
        next = group < n ? group : n;
        end = (n - next < 64u) ? n : next + 64u;
    }
    __device__ __forceinline__ bool empty() const { return next >= end; }
    // after handing out items: move to the wave's next group when the current one is used up
    __device__ __forceinline__ void take(uint32_t k) {
        next += k;
        if (next >= end && group < n) { group = (n - group <= stride) ? n : group + stride; open_group(); }
    }
};
// bid / nblk: this workgroup's index in, and the size of, the (virtual) grid that shares the n items.
__device__ __forceinline__ WavePool wave_pool(uint32_t n, uint32_t bid, uint32_t nblk) {
    WavePool wp;
    wp.n = n;
    wp.stride = nblk * RT_BLOCK;
    wp.group = (uint32_t)__builtin_amdgcn_readfirstlane((int)((bid * (RT_BLOCK / 64) + (threadIdx.x >> 6)) * 64u));
    wp.open_group();
    return wp;
}

__device__ __forceinline__ uint32_t lane_prefix(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// slab_pair with the ray kept as six scalars instead of six register pairs (the persistent kernels hold a ray's state
// across refills, and resident waves are what they are short of): the same (bound - o) * inv per component.
__device__ __forceinline__ void slab_pair_s(v3 o, v3 inv, float4 n0, float4 n1, float4 n2,
                                            float& te0, float& tx0, float& te1, float& tx1) {
    const float ax = (n0.x - o.x) * inv.x, ay = (n0.y - o.y) * inv.y, az = (n0.z - o.z) * inv.z;
    const float bx = (n0.w - o.x) * inv.x, by = (n1.x - o.y) * inv.y, bz = (n1.y - o.z) * inv.z;
    const float cx = (n1.z - o.x) * inv.x, cy = (n1.w - o.y) * inv.y, cz = (n2.x - o.z) * inv.z;
    const float dx = (n2.y - o.x) * inv.x, dy = (n2.z - o.y) * inv.y, dz = (n2.w - o.z) * inv.z;
    te0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    tx0 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    te1 = fmaxf(fmaxf(fminf(cx, dx), fminf(cy, dy)), fminf(cz, dz));
    tx1 = fminf(fminf(fmaxf(cx, dx), fmaxf(cy, dy)), fmaxf(cz, dz));
}

// One visit of a two-level node (WNode, rt_types.h): the slab test of slab_pair_s on each of its four slots, then the
// slots the ray passes sorted by entry distance.  r[k] = the k-th nearest passing slot's reference (RT_DONE from the
// first slot on that did not pass), t[k] its entry distance.  LIM: cull slots that start beyond `lim`.
template <bool LIM>
__device__ __forceinline__ void wide_visit(const WNode* __restrict__ wn, v3 o, v3 inv, float lim, int (&r)[4], float (&t)[4]) {
    const float4* np = reinterpret_cast<const float4*>(wn);
    const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4], n5 = np[5], n6 = np[6];
    const float INF = __int_as_float(0x7f800000);
#define RT_WSLOT(k, lx, ly, lz, hx, hy, hz, rf)                                                                  \
    {                                                                                                           \
        const float ax = ((lx) - o.x) * inv.x, ay = ((ly) - o.y) * inv.y, az = ((lz) - o.z) * inv.z;            \
        const float bx = ((hx) - o.x) * inv.x, by = ((hy) - o.y) * inv.y, bz = ((hz) - o.z) * inv.z;            \
        const float te = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));                             \
        const float tx = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));                             \
        const int ref = __float_as_int(rf);                                                                     \
        const bool p = tx >= te && tx >= 0.0f && (!LIM || !(te > lim)) && ref != -1;                            \
        r[k] = p ? ref : RT_DONE;                                                                               \
        t[k] = p ? fminf(te, 3.40282347e+38f) : INF;     /* a passing slot always sorts in front of the others */ \
    }
    RT_WSLOT(0, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n6.x)
    RT_WSLOT(1, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, n6.y)
    RT_WSLOT(2, n3.x, n3.y, n3.z, n3.w, n4.x, n4.y, n6.z)
    RT_WSLOT(3, n4.z, n4.w, n5.x, n5.y, n5.z, n5.w, n6.w)
#undef RT_WSLOT
#define RT_WCE(a, b) { const bool sw = t[b] < t[a]; const float ta = t[a]; const int ra = r[a];                  \
                       t[a] = sw ? t[b] : ta; r[a] = sw ? r[b] : ra; t[b] = sw ? ta : t[b]; r[b] = sw ? ra : r[b]; }
    RT_WCE(0, 1) RT_WCE(2, 3) RT_WCE(0, 2) RT_WCE(1, 3) RT_WCE(1, 2)
#undef RT_WCE
}

// Closest hit == lineCast (shader.comp:243-290) for the rays of `queue` (PRIMARY: the camera rays of all slots).
//
// Ray set-up (fetch, 1/d, root box) is done for a whole 64-ray group at a time with every lane of the wave working,
// whatever the lanes' own rays are doing, and the prepared rays wait in a per-wave LDS pool; a refill then costs an
// idle lane a few LDS reads.  (Setting rays up inside the refill, with only the idle lanes active, made every refill
// as expensive as a full wave's set-up.)
#define RT_POOL_WORDS 10     // o.xyz, d.xyz, 1/d.xyz, queue entry
// Below `nodeMin` lanes with a node to visit, the node phase of the refill kernels goes on until RT_TRI_MIN lanes have a
// leaf parked (so that the triangle phase runs with a fuller wave too) or fewer than RT_NODE_LOW lanes have a node
// (measured 16/12, 32/8, 48/4: 32/8 best on C2..C5).
#define RT_TRI_MIN 32
#define RT_NODE_LOW 8
// Scheduling knobs of the refill kernels, packed into one kernel argument (scalar registers, no VGPR):
//   bits 0..7 nodeMin, 8..15 triMin (0 = RT_TRI_MIN), 16..23 nodeLow (0 = RT_NODE_LOW), bit 24 defer: when the node phase
//   ends only because finished lanes can be refilled, the parked leaves wait (the refill comes first and the triangle phase
//   runs later with more lanes)
struct RefillSched { int nodeMin, triMin, nodeLow; bool defer; };
__device__ __forceinline__ RefillSched unpack_sched(int v) {
    RefillSched r;
    r.nodeMin = v & 255; r.triMin = (v >> 8) & 255; r.nodeLow = (v >> 16) & 255; r.defer = ((v >> 24) & 1) != 0;
    if (r.triMin == 0) r.triMin = RT_TRI_MIN;
    if (r.nodeLow == 0) r.nodeLow = RT_NODE_LOW;
    return r;
}
typedef float RayPool[RT_POOL_WORDS][64];      // one wave's pool of prepared rays
// WIDE: walk the two-level nodes (S.wnodes) instead of the binary ones — half the dependent fetches per ray.
template <bool COUNT, bool PRIMARY, bool WIDE = false>
__device__ __forceinline__ void extend_refill(const SceneView& S, const FrameParams& F, const PathBuffers& P,
                                              const float4* __restrict__ rays, const uint32_t* __restrict__ count,
                                              int rBase, uint32_t rayCount, int refillIdle, int schedBits,
                                              DeviceStats* stats, uint2* lds, RayPool* poolMem, uint32_t bid, uint32_t nblk) {
    const RefillSched sched = unpack_sched(schedBits);
    const int nodeMin = sched.nodeMin;
    const Stack stk = make_stack(lds, P);
    float (*pool)[64] = poolMem[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = *count;
    const float INF = __int_as_float(0x7f800000);
    WavePool wp = wave_pool(n, bid, nblk);
    uint32_t poolN = 0;          // prepared rays waiting in the pool (wave-uniform)
    uint32_t nNodes = 0, nTris = 0, maxSteps = 0, steps0 = 0;
    uint32_t wNode = 0, wTri = 0, bNodes = 0, bTris = 0;   // COUNT: wave-level step executions and the visits made in them (as k_extend's)
    uint32_t dbg2[4] = {0, 0, 0, 0};                       // COUNT: node / triangle steps made when no refill was possible, and their lanes
    uint32_t dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};            // COUNT, lane 0: node-phase exits [0], parked lanes at them [1], by reason [2..6], free lanes [7]

    bool active = false;
    uint32_t slot = 0;
    v3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 0.f);
    v3 rinv = mk(1.f, 1.f, 1.f);
    float bestT = INF, bestDist = 3.40282347e+38f, lim = INF;
    int bestIdx = -1, bestRef = -1;
    int sp = 0, cur = RT_DONE, pend = RT_DONE;   // pend: a leaf reached but not tested yet

    auto popNext = [&]() {
        cur = RT_DONE;
        while (sp > 0) {
            uint2 e = stk.pop(sp);
            if (__uint_as_float(e.y) > lim) continue;
            cur = (int)e.x;
            break;
        }
    };

    for (;;) {
        const unsigned long long idleMask = __ballot(!active);
        const int nIdle = __popcll(idleMask);
        if (nIdle >= refillIdle && (poolN > 0u || !wp.empty())) {
            if (poolN == 0u) {
                // set-up stage: lane l prepares ray wp.next + l of the wave's current group
                const uint32_t i = wp.next + lane;
                const bool have = i < wp.end;
                v3 ro = mk(0.f, 0.f, 0.f), rd = mk(1.f, 1.f, 1.f);
                bool ok = have;
                if (have) {
                    if (PRIMARY) {
                        int px, py, r;
                        ro = ld3(F.camPos);
                        ok = primary_ray(F, P, rBase, i, &px, &py, &r, &rd);
                        if (!ok) P.hit[i] = make_float2(0.f, __int_as_float(-2));
                    } else {
                        uint32_t fl, pathSlot;
                        unpack_ray(rays[2u * i], rays[2u * i + 1u], ro, rd, fl, pathSlot);
                    }
                }
                // Rays that could produce a NaN in a slab test (zero / non-finite direction component) go through the pool
                // too, marked in the top bit of their entry: the idle lane that receives one traces it on the spot with the
                // select-form routine (its own stack is empty then).
                bool ready = false, plain = true;
                v3 inv = mk(1.f, 1.f, 1.f);
                if (ok) {
                    plain = ray_is_plain(ro, rd);
                    ready = true;
                    if (plain) {
                        inv = mk(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                        float te;
                        if (COUNT) nNodes++;
                        if (!slab<true>(ro, inv, S.rootLo, S.rootHi, &te)) { P.hit[i] = make_float2(INF, __int_as_float(-1)); ready = false; }
                    }
                }
                const unsigned long long rm = __ballot(ready);
                if (ready) {
                    const uint32_t e = lane_prefix(rm);
                    pool[0][e] = ro.x; pool[1][e] = ro.y; pool[2][e] = ro.z;
                    pool[3][e] = rd.x; pool[4][e] = rd.y; pool[5][e] = rd.z;
                    pool[6][e] = inv.x; pool[7][e] = inv.y; pool[8][e] = inv.z;
                    pool[9][e] = __uint_as_float(plain ? i : (i | 0x80000000u));
                }
                poolN = (uint32_t)__popcll(rm);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                wp.take(wp.end - wp.next);
                continue;
            }
            // hand prepared rays to the idle lanes (from the top of the pool)
            const uint32_t pre = lane_prefix(idleMask);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (!active && pre < poolN) {
                const uint32_t e = poolN - 1u - pre;
                o = mk(pool[0][e], pool[1][e], pool[2][e]);
                d = mk(pool[3][e], pool[4][e], pool[5][e]);
                const v3 inv = mk(pool[6][e], pool[7][e], pool[8][e]);
                slot = __float_as_uint(pool[9][e]);
                if (slot & 0x80000000u) {
                    slot &= 0x7fffffffu;
                    HitRec best;
                    closest_hit<false, COUNT, false>(S, o, d, stk, best, nNodes, nTris);
                    P.hit[slot] = make_float2(best.t, __int_as_float(best.ref));
                } else {
                    rinv = inv;
                    bestT = INF; bestDist = 3.40282347e+38f; lim = INF; bestIdx = -1; bestRef = -1;
                    sp = 0; cur = S.rootChild; pend = RT_DONE;
                    if (cur < 0) { pend = cur; cur = RT_DONE; }      // the root's only child is a leaf
                    if (COUNT) steps0 = nNodes + nTris;
                    active = true;
                }
            }
            poolN -= ((uint32_t)nIdle < poolN) ? (uint32_t)nIdle : poolN;
            __builtin_amdgcn_wave_barrier();
            if (poolN == 0u && !wp.empty()) continue;      // still idle lanes to feed: prepare the next group first
        } else if (nIdle == 64) break;
        const bool canRefill = poolN > 0u || !wp.empty();

        for (;;) {
            // Node phase.  A wave-level load instruction costs the same however few lanes use it, so node
            // steps run while at least `nodeMin` lanes have a node to visit.  A lane that reaches a leaf parks it in
            // `pend` and goes on with the next node of its stack; with a leaf parked and another at hand it waits.  The
            // phase ends when too few lanes are left and there is something else to do (parked leaves to test, or
            // finished lanes to refill).
            bool skipTris = false;
            for (;;) {
                const bool hasNode = cur >= 0;
                const int nNode = __popcll(__ballot(hasNode));
                if (nNode < nodeMin) {
                    const int nParked = __popcll(__ballot(pend != RT_DONE));
                    const int nFree = __popcll(__ballot(!active || (cur == RT_DONE && pend == RT_DONE)));
                    const bool wantTris = nParked >= sched.triMin || (nParked > 0 && nNode < sched.nodeLow);
                    const bool wantRefill = canRefill && nFree >= refillIdle;
                    if (wantRefill && sched.defer && nParked < sched.triMin) { skipTris = true; if (COUNT && lane == 0) dbg[4]++; break; }
                    if (nNode == 0 || wantTris || wantRefill) {
                        if (COUNT && lane == 0) {
                            dbg[0]++; dbg[1] += (uint32_t)nParked;
                            if (nParked >= sched.triMin) dbg[2]++; else if (wantTris) dbg[3]++; else if (wantRefill) dbg[5]++; else dbg[6]++;
                            dbg[7] += (uint32_t)nFree;
                        }
                        break;
                    }
                }
                if (hasNode && WIDE) {
                    // (a wide visit counts as two node visits: it fetches two node records' worth, 128 bytes)
                    if (COUNT) { nNodes += 2; bNodes += 2; if (first_active_lane()) { wNode += 2; if (!canRefill) { dbg2[0]++; dbg2[1] += (uint32_t)nNode; } } }
                    int r[4]; float t[4];
                    wide_visit<true>(S.wnodes + cur, o, rinv, lim, r, t);
                    for (int k = 3; k >= 1; k--)      // the farther slots wait on the stack, nearest on top
                        if (r[k] != RT_DONE) { uint2 e; e.x = (uint32_t)r[k]; e.y = __float_as_uint(t[k]); stk.push(sp, e); }
                    if (r[0] != RT_DONE) cur = r[0]; else popNext();
                    if (cur < 0 && cur != RT_DONE && pend == RT_DONE) { pend = cur; popNext(); }
                } else if (hasNode) {
                    const float4* np = reinterpret_cast<const float4*>(S.nodes + cur);
                    float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                    if (COUNT) { nNodes++; bNodes++; if (first_active_lane()) { wNode++; if (!canRefill) { dbg2[0]++; dbg2[1] += (uint32_t)nNode; } } }
                    float te0, tx0, te1, tx1;
                    slab_pair_s(o, rinv, n0, n1, n2, te0, tx0, te1, tx1);
                    bool p0 = tx0 >= te0 && tx0 >= 0.0f && !(te0 > lim);
                    bool p1 = tx1 >= te1 && tx1 >= 0.0f && !(te1 > lim);
                    int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
                    if (p0 && p1) {
                        bool swap = te1 < te0;
                        uint2 e; e.x = (uint32_t)(swap ? c0 : c1); e.y = __float_as_uint(swap ? te0 : te1);
                        stk.push(sp, e);
                        cur = swap ? c1 : c0;
                    } else if (p0) cur = c0;
                    else if (p1) cur = c1;
                    else popNext();
                    if (cur < 0 && cur != RT_DONE && pend == RT_DONE) { pend = cur; popNext(); }
                }
            }
            // Triangle phase: every lane with a parked leaf tests it; a second leaf waiting in `cur` is parked next.
            // (One triangle per phase instead of the whole leaf, so that a phase is as short as its shortest leaf, was
            // measured: 4 % slower.)
            if (!skipTris && pend != RT_DONE) {
                int code = -pend - 1;
                int first = code >> 3, cnt = code & 7;
                const float4* tp = reinterpret_cast<const float4*>(S.tris + first);
                for (int k = 0; k < cnt; k++) {
                    if (COUNT) { nTris++; bTris++; if (first_active_lane()) { wTri++; if (!canRefill) { dbg2[2]++; dbg2[3] += (uint32_t)__popcll(__ballot(true)); } } }
                    float4 A = tp[3 * k], B = tp[3 * k + 1], C = tp[3 * k + 2];
                    float t;
                    if (tri_t(A, B, C, o, d, &t)) {
                        v3 h = o + d * t;
                        v3 oh = o - h;
                        float newDist = dot3(oh, oh);
                        int idx = first + k;
                        if (newDist < bestDist || (newDist == bestDist && idx < bestIdx)) {
                            bestDist = newDist; bestT = t; bestIdx = idx; bestRef = __float_as_int(C.y);
                            lim = bestT + bestT * 1e-4f + S.cullAbs;
                        }
                    }
                }
                pend = RT_DONE;
                if (cur < 0 && cur != RT_DONE) { pend = cur; popNext(); }
            }
            if (active && cur == RT_DONE && pend == RT_DONE) {
                P.hit[slot] = make_float2(bestT, __int_as_float(bestRef));
                if (COUNT) { uint32_t st = nNodes + nTris - steps0 + 1; if (st > maxSteps) maxSteps = st; }
                active = false;
            }
            const unsigned long long am = __ballot(active);
            if (am == 0ull) break;
            if (canRefill && 64 - __popcll(am) >= refillIdle) break;
        }
    }

    if (bid == 0 && threadIdx.x == 0)
        atomicAdd(PRIMARY ? &stats->rays_primary : &stats->rays_bounce, (unsigned long long)(rayCount ? rayCount : n));
    if (COUNT) {
        __shared__ uint32_t red[6];
        if (threadIdx.x < 6) red[threadIdx.x] = 0;
        __syncthreads();
        for (int off = 32; off > 0; off >>= 1) {
            nNodes += __shfl_down(nNodes, off, 64); nTris += __shfl_down(nTris, off, 64);
            bNodes += __shfl_down(bNodes, off, 64); bTris += __shfl_down(bTris, off, 64);
            wNode += __shfl_down(wNode, off, 64); wTri += __shfl_down(wTri, off, 64);
        }
        for (int off = 32; off > 0; off >>= 1) { uint32_t o2 = __shfl_down(maxSteps, off, 64); if (o2 > maxSteps) maxSteps = o2; }
        for (int k = 0; k < 4; k++) if (dbg2[k]) atomicAdd(&stats->dbg2[k], (unsigned long long)dbg2[k]);
        if ((threadIdx.x & 63) == 0) {
            for (int k = 0; k < 8; k++) if (dbg[k]) atomicAdd(&stats->dbg[k], (unsigned long long)dbg[k]);

            atomicAdd(&red[0], nNodes); atomicAdd(&red[1], nTris); atomicAdd(&red[2], bNodes); atomicAdd(&red[3], bTris);
            atomicAdd(&red[4], wNode); atomicAdd(&red[5], wTri);
            if (maxSteps) atomicMax(&stats->max_ray_steps, (unsigned long long)maxSteps);
        }
        __syncthreads();
        if (threadIdx.x == 0 && (red[0] | red[1])) {
            atomicAdd(&stats->node_visits, (unsigned long long)red[0]);
            atomicAdd(&stats->tri_tests, (unsigned long long)red[1]);
            if (!PRIMARY) {
                atomicAdd(&stats->bounce_node_visits, (unsigned long long)red[2]);
                atomicAdd(&stats->bounce_tri_tests, (unsigned long long)red[3]);
                atomicAdd(&stats->wave_node_steps, (unsigned long long)red[4]);
                atomicAdd(&stats->wave_tri_steps, (unsigned long long)red[5]);
            }
        }
    }
}

template <bool COUNT, bool PRIMARY, bool WIDE = false>
__global__ __launch_bounds__(RT_BLOCK, (WIDE ? 5 : 6)) void k_extend_refill(SceneView S, FrameParams F, PathBuffers P,
                                                         const float4* __restrict__ rays, const uint32_t* __restrict__ count,
                                                         int rBase, uint32_t rayCount, int refillIdle, int schedBits,
                                                         DeviceStats* stats) {
    __shared__ uint2 lds[RT_LDS_STACK * RT_BLOCK];
    __shared__ RayPool poolMem[RT_BLOCK / 64];
    extend_refill<COUNT, PRIMARY, WIDE>(S, F, P, rays, count, rBase, rayCount, refillIdle, schedBits, stats, lds, poolMem, blockIdx.x, gridDim.x);
}

// shader.comp:453-484: the shadow lineCast of every log entry awaiting one; the entry becomes "add" (2) or "none" (0).
// Work item i = (shadow-queue entry i / numLights, light i % numLights).
// SUN1: exactly one light and it is a sun — no point-light state is carried (fewer registers, more resident waves, which is
// what a miss-latency-bound kernel wants) and the direction's reciprocal is computed once per thread.
template <bool COUNT, bool SUN1, bool WIDE = false>
__device__ __forceinline__ void shadow_refill(const SceneView& S, const FrameParams& F, const PathBuffers& P, const uint32_t* __restrict__ count,
                                              int bounce, int refillIdle, int schedBits, DeviceStats* stats, uint2* lds, uint32_t bid, uint32_t nblk) {
    const Stack stk = make_stack(lds, P);
    const RefillSched sched = unpack_sched(schedBits);
    const int nodeMin = sched.nodeMin;
    const uint32_t L = (uint32_t)F.numLights;
    const uint32_t n = *count * L;
    const float INF = __int_as_float(0x7f800000);
    WavePool wp = wave_pool(n, bid, nblk);
    uint32_t nNodes = 0, nTris = 0;

    bool active = false, occ = false;
    // the ray's log entry, zeroed if the light turns out to be occluded (PathBuffers::contrib).  SUN1: the entry's index in
    // the bounce's plane instead of its address (one register less: the fused kernel of the two-level walk sits at the
    // 80-register step between 6 and 5 waves per SIMD)
    LogEntry* wptr = nullptr;
    uint32_t wslot = 0;
    LogEntry* const sunPlane = SUN1 ? log_plane(P, bounce).base : nullptr;
    v3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 0.f), shadeLoc = mk(0.f, 0.f, 0.f);
    v3 rinv = mk(1.f, 1.f, 1.f);
    float distToLight = 0.f, lim = INF;
    int kind = 1;
    int sp = 0, cur = RT_DONE, pend = RT_DONE;     // pend: a leaf reached but not tested yet
    auto popS = [&]() { cur = sp > 0 ? (int)stk.pop(sp).x : RT_DONE; };
    v3 sunInv = mk(1.f, 1.f, 1.f);
    bool sunPlain = true;
    if (SUN1) {
        const RefLight& Lg = F.lights[0];
        d = -mk(Lg.dir.x, Lg.dir.y, Lg.dir.z);
        sunInv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        sunPlain = ray_is_plain(mk(0.f, 0.f, 0.f), d);
    }

    for (;;) {
        const unsigned long long idleMask = __ballot(!active);
        const int nIdle = __popcll(idleMask);
        if (!wp.empty() && nIdle >= refillIdle) {
            {
                const uint32_t pre = lane_prefix(idleMask);
                const uint32_t avail = wp.end - wp.next;
                if (!active && pre < avail) {
                    const uint32_t i = wp.next + pre;
                    const uint32_t q = (L == 1u) ? i : i / L, l = (L == 1u) ? 0u : i % L;
                    const float4 rec = P.shadowRec[q];
                    const uint32_t slot = __float_as_uint(rec.w);      // the path's log index (log_plane)
                    const LogPlane LP = log_plane(P, bounce);
                    LogEntry* wp_ = SUN1 ? nullptr : &LP.base[(size_t)l * LP.stride + slot];
                    // one light: every queued entry awaits its shadow ray (k_shade queued it for that reason), so nothing
                    // of the entry is read here; it is only overwritten (with zeros) if the light turns out to be occluded
                    if (SUN1) {
                        o = mk(rec.x, rec.y, rec.z);
                        wslot = slot;
                        if (!(sunPlain && ray_is_plain(o, mk(1.f, 1.f, 1.f)))) {
                            bool oc = occluded_fast<COUNT, false>(S, o, d, 1, mk(0.f, 0.f, 0.f), 0.f, stk, nNodes, nTris);
                            if (oc) sunPlane[wslot] = log_entry(0.f, 0.f, 0.f);
                        } else {
                            float te;
                            if (COUNT) nNodes++;
                            if (slab<true>(o, sunInv, S.rootLo, S.rootHi, &te)) {      // (a ray that misses the scene's box is not occluded: the entry stands)
                                rinv = sunInv;
                                occ = false; sp = 0; cur = S.rootChild; pend = RT_DONE;
                                if (cur < 0) { pend = cur; cur = RT_DONE; }
                                active = true;
                            }
                        }
                    } else {
                        float4 L0, L1;
                        if (L == 1u) {         // kernarg, uniform: scalar loads
                            const RefLight& Lg = F.lights[0];
                            L0 = make_float4(Lg.loc.x, Lg.loc.y, Lg.loc.z, Lg.strength);
                            L1 = make_float4(Lg.dir.x, Lg.dir.y, Lg.dir.z, __uint_as_float(Lg.lightType));
                        } else {
                            const float4* lp = reinterpret_cast<const float4*>(S.lights + l);
                            L0 = lp[0]; L1 = lp[1];
                        }
                        // several lights: the entries that await a shadow ray are those of a light type the shader casts
                        // for whose contribution is not exactly zero (shadow_batch)
                        bool awaits = true;
                        if (L != 1u) {
                            const LogEntry e = *wp_;
                            awaits = __float_as_uint(L1.w) <= 1u && !(e.x == 0.f && e.y == 0.f && e.z == 0.f);
                        }
                        if (awaits) {
                            o = mk(rec.x, rec.y, rec.z); shadeLoc = mk(0.f, 0.f, 0.f);
                            if (F.anyPoint) { float4 h4 = P.shadeLoc[slot]; shadeLoc = mk(h4.x, h4.y, h4.z); }
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
                                if (oc) *wptr = log_entry(0.f, 0.f, 0.f);
                            } else {
                                v3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                                float te;
                                if (COUNT) nNodes++;
                                if (slab<true>(o, inv, S.rootLo, S.rootHi, &te)) {
                                    rinv = inv;
                                    lim = (kind == 0) ? (distToLight + distToLight * 1e-3f + S.cullAbs * 4.0f) : INF;
                                    occ = false; sp = 0; cur = S.rootChild; pend = RT_DONE;
                                    if (cur < 0) { pend = cur; cur = RT_DONE; }
                                    active = true;
                                }
                            }
                        }
                    }
                }
                wp.take(((uint32_t)nIdle < avail) ? (uint32_t)nIdle : avail);
                continue;
            }
        }
        if (nIdle == 64) break;
        const bool canRefill = !wp.empty();

        for (;;) {
            // node phase / triangle phase with parked leaves, as in k_extend_refill
            bool skipTris = false;
            for (;;) {
                const bool hasNode = cur >= 0;
                const int nNode = __popcll(__ballot(hasNode));
                if (nNode < nodeMin) {
                    const int nParked = __popcll(__ballot(pend != RT_DONE));
                    const int nFree = __popcll(__ballot(!active || (cur == RT_DONE && pend == RT_DONE)));
                    const bool wantTris = nParked >= sched.triMin || (nParked > 0 && nNode < sched.nodeLow);
                    const bool wantRefill = canRefill && nFree >= refillIdle;
                    if (wantRefill && sched.defer && nParked < sched.triMin) { skipTris = true; break; }
                    if (nNode == 0 || wantTris || wantRefill) break;
                }
                if (hasNode && WIDE) {
                    if (COUNT) nNodes += 2;
                    int r[4]; float t[4];
                    if (SUN1) wide_visit<false>(S.wnodes + cur, o, rinv, 0.f, r, t); else wide_visit<true>(S.wnodes + cur, o, rinv, lim, r, t);
                    for (int k = 3; k >= 1; k--)      // nearer slots first: an occluder, if any, tends to be found sooner
                        if (r[k] != RT_DONE) { uint2 e; e.x = (uint32_t)r[k]; e.y = 0u; stk.push(sp, e); }
                    if (r[0] != RT_DONE) cur = r[0]; else popS();
                    if (cur < 0 && cur != RT_DONE && pend == RT_DONE) { pend = cur; popS(); }
                } else if (hasNode) {
                    const float4* np = reinterpret_cast<const float4*>(S.nodes + cur);
                    float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                    if (COUNT) nNodes++;
                    float te0, tx0, te1, tx1;
                    slab_pair_s(o, rinv, n0, n1, n2, te0, tx0, te1, tx1);
                    bool p0 = tx0 >= te0 && tx0 >= 0.0f && (SUN1 || !(te0 > lim));
                    bool p1 = tx1 >= te1 && tx1 >= 0.0f && (SUN1 || !(te1 > lim));
                    int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
                    if (p0 && p1) {      // nearer box first: an occluder, if any, tends to be found sooner
                        bool swap = te1 < te0;
                        uint2 e; e.x = (uint32_t)(swap ? c0 : c1); e.y = 0u; stk.push(sp, e);
                        cur = swap ? c1 : c0;
                    } else if (p0) cur = c0;
                    else if (p1) cur = c1;
                    else popS();
                    if (cur < 0 && cur != RT_DONE && pend == RT_DONE) { pend = cur; popS(); }
                }
            }
            if (!skipTris && pend != RT_DONE) {
                int code = -pend - 1;
                int first = code >> 3, cnt = code & 7;
                const float4* tp = reinterpret_cast<const float4*>(S.tris + first);
                for (int k = 0; k < cnt; k++) {
                    if (COUNT) nTris++;
                    float t;
                    if (tri_t(tp[3 * k], tp[3 * k + 1], tp[3 * k + 2], o, d, &t)) {
                        if (SUN1 || kind == 1) occ = true;
                        else { v3 h = o + d * t; v3 dl = h - shadeLoc; if (sqrtf(dot3(dl, dl)) < distToLight) occ = true; }
                    }
                }
                pend = RT_DONE;
                if (occ) { cur = RT_DONE; sp = 0; }
                else if (cur < 0 && cur != RT_DONE) { pend = cur; popS(); }
            }
            if (active && cur == RT_DONE && pend == RT_DONE) {
                if (occ) { if (SUN1) sunPlane[wslot] = log_entry(0.f, 0.f, 0.f); else *wptr = log_entry(0.f, 0.f, 0.f); }
                active = false;
            }
            const unsigned long long am = __ballot(active);
            if (am == 0ull) break;
            if (canRefill && 64 - __popcll(am) >= refillIdle) break;
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

template <bool COUNT, bool SUN1, bool WIDE = false>
__global__ __launch_bounds__(RT_BLOCK) void k_shadow_refill(SceneView S, FrameParams F, PathBuffers P, const uint32_t* __restrict__ count,
                                                         int bounce, int refillIdle, int schedBits, DeviceStats* stats) {
    __shared__ uint2 lds[RT_LDS_STACK * RT_BLOCK];
    shadow_refill<COUNT, SUN1, WIDE>(S, F, P, count, bounce, refillIdle, schedBits, stats, lds, blockIdx.x, gridDim.x);
}

// The shadow rays of bounce b and the closest-hit rays of bounce b + 1 in ONE launch.  Both queues are written by
// k_shade of bounce b and neither kernel reads what the other writes, yet on a stream they ran one after the other — and
// late bounces are short queues of long rays (scene 7: 75 k .. 800 k rays from bounce 4 on, up to ~750 traversal steps
// each), whose launches last as long as their slowest ray with the chip almost idle.  Run together, a bounce costs the
// slower of the two instead of their sum, and on large queues a wave that has finished its share of one kind goes
// straight on with the other instead of idling through the first kernel's drain.
// Workgroup b of the launch runs the closest-hit body as workgroup b of a grid of gE, then the shadow body as workgroup
// (gridDim - 1 - b) of a grid of gS: short queues therefore land on opposite ends of the launch and run side by side.
// EXT: 0 extend_batch, 1 extend_refill.  SHD: 0 shadow_batch, 1 shadow_sun_batch, 2 shadow_refill<., false>, 3 shadow_refill<., true>,
// 4 shadow_sun_packet (coherent rays: bounce 0).
// WIDE: the refill bodies walk the two-level nodes.
template <int EXT, int SHD, bool WIDE = false>
__global__ __launch_bounds__(RT_BLOCK, ((EXT == 1 && SHD != 2) ? (WIDE ? 5 : 6) : 1)) void k_trace_pair(SceneView S, FrameParams F, PathBuffers P,
                                                                          const float4* __restrict__ rays, const uint32_t* __restrict__ countExt,
                                                                          int rBase, const uint32_t* __restrict__ countShd, int bounceShd,
                                                                          uint32_t gE, uint32_t gS, int refillIdle, int schedBits, DeviceStats* stats) {
    __shared__ uint2 lds[RT_LDS_STACK * RT_BLOCK];
    __shared__ RayPool poolMem[EXT == 1 ? RT_BLOCK / 64 : 1];
    if (blockIdx.x < gE) {
        if (EXT == 1) extend_refill<false, false, WIDE>(S, F, P, rays, countExt, rBase, 0u, refillIdle, schedBits, stats, lds, poolMem, blockIdx.x, gE);
        else extend_batch<false, false, false>(S, F, P, rays, countExt, rBase, 0u, stats, lds, blockIdx.x, gE);
    }
    const uint32_t bs = gridDim.x - 1u - blockIdx.x;
    if (bs < gS) {
        if (SHD == 0) shadow_batch<false, false>(S, F, P, countShd, bounceShd, stats, lds, bs, gS);
        else if (SHD == 1) shadow_sun_batch(S, F, P, countShd, bounceShd, lds, bs, gS);
        else if (SHD == 2) shadow_refill<false, false, WIDE>(S, F, P, countShd, bounceShd, refillIdle, schedBits, stats, lds, bs, gS);
        else if (SHD == 3) shadow_refill<false, true, WIDE>(S, F, P, countShd, bounceShd, refillIdle, schedBits, stats, lds, bs, gS);
        else shadow_sun_packet(S, F, P, countShd, bounceShd, lds, bs, gS);
    }
}

}  // namespace rt

#include "rt_trace_packet.hip.h"
