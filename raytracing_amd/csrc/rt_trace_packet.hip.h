// gfx950 packet traversal for coherent rays (included by rt_kernels.hip.h after rt_trace_refill.hip.h).
//
// The shadow rays of bounce 0 towards a sun are the most coherent rays of a frame: one direction, and 64 consecutive
// entries of the shadow queue start at the camera-ray hits of a few neighbouring pixels (slot_split: 16 samples of 4 pixels,
// or 64 samples of one).  The per-lane kernels still pay a wave-level vector load per 16 bytes of every node they visit
// (>= 17 CU-cycles each however the addresses coincide).  Here a WAVE walks the tree once for its 64 rays:
//   * the node and triangle records are read with SCALAR loads (wave-uniform address, scalar data cache, no texture
//     addresser work at all) and the slab / triangle arithmetic takes them as scalar operands;
//   * control flow is scalar: the traversal stack holds (node, 64-bit lane mask) entries and lives in the lanes of three
//     registers (entry k in lane k: written by a select on the lane index, read with v_readlane at a uniform index) — no LDS, no memory;
//   * a lane takes part in a node only if its own ray passed every box on the way to it, so per ray the set of triangles
//     tested is the per-lane walk's: a ray is occluded iff occluded_fast<.., true>(kind 1) says so (any-hit: the order
//     of the tests does not matter).  Same float expressions as slab<true> / tri_t (shader.comp:104-125, 171-184).
// Packets in which some ray could produce a NaN in a slab test take the per-lane select-form routine, as everywhere.
#pragma once

namespace rt {

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 64 / 16 bytes through the scalar data cache (read only), complete on return
__device__ __forceinline__ u32x16 sload16(const void* p) {
    u32x16 v;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
__device__ __forceinline__ void sload_tri(const void* p, u32x4& a, u32x4& b, u32x4& c) {
    asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_load_dwordx4 %2, %3, 0x20\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c) : "s"(p) : "memory");
}
__device__ __forceinline__ float uf(uint32_t x) { return __uint_as_float(x); }
__device__ __forceinline__ unsigned long long read_mask(int lo, int hi, int k) {
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(lo, k) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(hi, k) << 32);
}

// shader.comp:453-484 for the queued shadow rays of a one-sun scene, a packet of 64 consecutive queue entries per wave
// step.  stkFallback: the per-lane stack for packets that hold a ray the select-form routine has to trace.
__device__ __forceinline__ void shadow_sun_packet(const SceneView& S, const FrameParams& F, const PathBuffers& P, const uint32_t* __restrict__ count,
                                                  int bounce, uint2* lds, uint32_t bid, uint32_t nblk) {
    const Stack stkFallback = make_stack(lds, P);
    const uint32_t n = *count;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long laneBit = 1ull << lane;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(bid * (RT_BLOCK / 64) + (threadIdx.x >> 6)));
    const uint32_t nWaves = nblk * (RT_BLOCK / 64);
    const RefLight Lg = F.lights[0];
    const v3 d = -mk(Lg.dir.x, Lg.dir.y, Lg.dir.z);
    const v3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const float big = 3.0e38f;
    const bool dPlain = fabsf(d.x) < big && fabsf(d.y) < big && fabsf(d.z) < big && d.x != 0.f && d.y != 0.f && d.z != 0.f;
    LogEntry* const logBase = log_plane(P, bounce).base;
    const uint32_t nGroups = (n + 63u) / 64u;
    for (uint32_t g = wave; g < nGroups; g += nWaves) {
        const uint32_t i = g * 64u + lane;
        const bool have = i < n;
        float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have) o4 = P.shadowRec[i];
        const uint32_t slot = __float_as_uint(o4.w);
        const v3 o = mk(o4.x, o4.y, o4.z);
        const bool plain = dPlain && fabsf(o.x) < big && fabsf(o.y) < big && fabsf(o.z) < big;      // == ray_is_plain(o, d)
        bool occ = false;
        if (__ballot(have && !plain) != 0ull) {
            uint32_t nNodes = 0, nTris = 0;
            if (have) occ = plain ? occluded_fast<false, true>(S, o, d, 1, mk(0.f, 0.f, 0.f), 0.f, stkFallback, nNodes, nTris)
                                  : occluded_fast<false, false>(S, o, d, 1, mk(0.f, 0.f, 0.f), 0.f, stkFallback, nNodes, nTris);
        } else {
            float te;
            const bool in = have && slab<true>(o, inv, S.rootLo, S.rootHi, &te);
            unsigned long long alive = __ballot(in);          // rays that are in the walk and not yet occluded
            if (alive != 0ull) {
                int stNode = 0, stLo = 0, stHi = 0;             // the stack: entry k in lane k of these
                int sp = 0;
                int cur = S.rootChild;
                unsigned long long cm = alive;                  // the lanes whose rays passed every box down to `cur`
                for (;;) {
                    bool popNow = false;
                    if (cur >= 0) {
                        const u32x16 nd = sload16(S.nodes + cur);
                        bool p0 = false, p1 = false, sw = false;
                        if ((cm & laneBit) != 0ull) {
                            // slab<true> on both child boxes (slab_pair_s), the bounds as scalar operands
                            const float ax = (uf(nd[0]) - o.x) * inv.x, ay = (uf(nd[1]) - o.y) * inv.y, az = (uf(nd[2]) - o.z) * inv.z;
                            const float bx = (uf(nd[3]) - o.x) * inv.x, by = (uf(nd[4]) - o.y) * inv.y, bz = (uf(nd[5]) - o.z) * inv.z;
                            const float cx = (uf(nd[6]) - o.x) * inv.x, cy = (uf(nd[7]) - o.y) * inv.y, cz = (uf(nd[8]) - o.z) * inv.z;
                            const float dx = (uf(nd[9]) - o.x) * inv.x, dy = (uf(nd[10]) - o.y) * inv.y, dz = (uf(nd[11]) - o.z) * inv.z;
                            const float te0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
                            const float tx0 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
                            const float te1 = fmaxf(fmaxf(fminf(cx, dx), fminf(cy, dy)), fminf(cz, dz));
                            const float tx1 = fminf(fminf(fmaxf(cx, dx), fmaxf(cy, dy)), fmaxf(cz, dz));
                            p0 = tx0 >= te0 && tx0 >= 0.0f;
                            p1 = tx1 >= te1 && tx1 >= 0.0f;
                            sw = te1 < te0;
                        }
                        const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1);
                        const int c0 = (int)nd[12], c1 = (int)nd[13];
                        if (m0 != 0ull && m1 != 0ull) {
                            // the nearer box first, as the first ray that passes both sees it (any order gives the same verdicts)
                            const unsigned long long both = m0 & m1;
                            const unsigned long long swm = __ballot(sw);
                            const bool swap = both != 0ull && ((swm >> __builtin_ctzll(both)) & 1ull) != 0ull;
                            const int farC = swap ? c0 : c1;
                            const unsigned long long farM = swap ? m0 : m1;
                            const bool top = (int)lane == sp;      // (a select per register: entry sp lives in lane sp)
                            stNode = top ? farC : stNode;
                            stLo = top ? (int)(uint32_t)farM : stLo;
                            stHi = top ? (int)(uint32_t)(farM >> 32) : stHi;
                            sp++;
                            cur = swap ? c1 : c0; cm = swap ? m1 : m0;
                        } else if (m0 != 0ull) { cur = c0; cm = m0; }
                        else if (m1 != 0ull) { cur = c1; cm = m1; }
                        else popNow = true;
                    } else {
                        const int code = -cur - 1;
                        const int first = code >> 3, cnt = code & 7;
                        for (int k = 0; k < cnt; k++) {
                            u32x4 ta, tb, tc;
                            sload_tri(S.tris + first + k, ta, tb, tc);
                            if ((cm & laneBit) != 0ull) {
                                float t;
                                if (tri_t(make_float4(uf(ta[0]), uf(ta[1]), uf(ta[2]), uf(ta[3])), make_float4(uf(tb[0]), uf(tb[1]), uf(tb[2]), uf(tb[3])),
                                          make_float4(uf(tc[0]), uf(tc[1]), uf(tc[2]), uf(tc[3])), o, d, &t)) occ = true;
                            }
                        }
                        alive &= ~__ballot(occ);
                        if (alive == 0ull) break;
                        popNow = true;
                    }
                    if (popNow) {
                        bool found = false;
                        while (sp > 0) {
                            sp--;
                            const unsigned long long m = read_mask(stLo, stHi, sp) & alive;
                            if (m != 0ull) { cur = __builtin_amdgcn_readlane(stNode, sp); cm = m; found = true; break; }
                        }
                        if (!found) break;
                    }
                }
            }
        }
        if (have && occ) logBase[slot] = log_entry(0.f, 0.f, 0.f);      // an occluded light adds nothing (PathBuffers::contrib)
    }
}

__global__ __launch_bounds__(RT_BLOCK) void k_shadow_sun_packet(SceneView S, FrameParams F, PathBuffers P, const uint32_t* __restrict__ count, int bounce) {
    __shared__ uint2 lds[RT_LDS_STACK * RT_BLOCK];
    shadow_sun_packet(S, F, P, count, bounce, lds, blockIdx.x, gridDim.x);
}

}  // namespace rt
