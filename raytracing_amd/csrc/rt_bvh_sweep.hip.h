// Device side of the SAH builder's sweep (included by rt_core.hip; registered with csrc/host/bvh_build.h).
//
// Model::computeForAxis (/root/reference/Raytracing/model.cpp:425-461) after its sort: inclusive prefix and suffix
// boxes over the node's triangles in sorted order, cost(i) = 1 + SA(prefix_i) / SA(all) * i + SA(suffix_i) / SA(all) * (n - i)
// in float32, first strict minimum.  The host keeps the sorts (std::sort's order among equal keys is part of the tree and
// no device sort reproduces it) and, for nodes of 2^18 triangles and more, sends the sorted index list here: at 20 M
// triangles a sweep is 80 MB over PCIe and a few milliseconds of kernels instead of ~0.4 s of one host core.
// Exactness: min / max of non-NaN floats are exact and order-independent up to the sign of a zero, which area()'s
// differences absorb (bvh_build.h); cost is evaluated per element with the reference's expression, IEEE division,
// no contraction (-ffp-contract=off).  Scenes with a NaN bound never come here.
#pragma once
#include <mutex>
#include "host/bvh_build.h"

namespace rt {
namespace sweep {

using rt_host::bvh_detail::B6;

constexpr int kPerThread = 8;
constexpr int kTile = 256 * kPerThread;

struct Box { float lx, ly, lz, hx, hy, hz; };
static_assert(sizeof(Box) == sizeof(B6), "Box mirrors the builder's B6");
__device__ __forceinline__ Box box_identity() { const float inf = __int_as_float(0x7f800000); return {inf, inf, inf, -inf, -inf, -inf}; }
// glm::min(x, y) = y < x ? y : x, glm::max(x, y) = x < y ? y : x (model.cpp:433-442)
__device__ __forceinline__ Box box_union(const Box& a, const Box& b) {
    return {b.lx < a.lx ? b.lx : a.lx, b.ly < a.ly ? b.ly : a.ly, b.lz < a.lz ? b.lz : a.lz,
            a.hx < b.hx ? b.hx : a.hx, a.hy < b.hy ? b.hy : a.hy, a.hz < b.hz ? b.hz : a.hz};
}
__device__ __forceinline__ float box_area(const Box& b) {      // model.h:115-118
    float ex = b.hx - b.lx, ey = b.hy - b.ly, ez = b.hz - b.lz;
    return 2.f * (ex * ey + ey * ez + ez * ex);
}

// union of the block's 256 per-thread boxes into lds[0] (tree reduction)
__device__ __forceinline__ Box block_union(Box v, Box* lds) {
    lds[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) lds[threadIdx.x] = box_union(lds[threadIdx.x], lds[threadIdx.x + s]);
        __syncthreads();
    }
    Box r = lds[0];
    __syncthreads();
    return r;
}

// pass 1: sorted[i] = boxes[ids[i]] (contiguous copy in sorted order) and every tile's union
__global__ __launch_bounds__(256) void k_sweep_gather(const Box* __restrict__ boxes, const uint32_t* __restrict__ ids, uint32_t n,
                                                      Box* __restrict__ sorted, Box* __restrict__ tileTotal) {
    __shared__ Box lds[256];
    const uint32_t base = blockIdx.x * kTile;
    Box t = box_identity();
    for (int k = 0; k < kPerThread; k++) {
        const uint32_t i = base + (uint32_t)k * 256u + threadIdx.x;          // coalesced over the tile
        if (i < n) { const Box b = boxes[ids[i]]; sorted[i] = b; t = box_union(t, b); }
    }
    t = block_union(t, lds);
    if (threadIdx.x == 0) tileTotal[blockIdx.x] = t;
}

// pass 2: exclusive prefix (thread 0) and exclusive suffix (thread 1) of the tile unions; a few thousand tiles at most
__global__ void k_sweep_tiles(const Box* __restrict__ tileTotal, uint32_t nTiles, Box* __restrict__ tileLeft, Box* __restrict__ tileRight) {
    if (threadIdx.x == 0) {
        Box acc = box_identity();
        for (uint32_t t = 0; t < nTiles; t++) { tileLeft[t] = acc; acc = box_union(acc, tileTotal[t]); }
    } else if (threadIdx.x == 1) {
        Box acc = box_identity();
        for (uint32_t t = nTiles; t-- > 0;) { tileRight[t] = acc; acc = box_union(acc, tileTotal[t]); }
    }
}

// pass 3: per element the inclusive prefix / suffix boxes and the cost; per tile the first minimum below FLT_MAX
__global__ __launch_bounds__(256) void k_sweep_cost(const Box* __restrict__ sorted, uint32_t n, const Box* __restrict__ tileLeft,
                                                    const Box* __restrict__ tileRight, float* __restrict__ tileCost, int* __restrict__ tileIndex) {
    __shared__ Box ldsL[256], ldsR[256];
    __shared__ float ldsC[256];
    __shared__ int ldsI[256];
    const uint32_t first = blockIdx.x * kTile + threadIdx.x * kPerThread;     // this thread's run of kPerThread consecutive elements
    Box e[kPerThread];
    Box mine = box_identity();
    for (int k = 0; k < kPerThread; k++) {
        e[k] = (first + k < n) ? sorted[first + k] : box_identity();
        mine = box_union(mine, e[k]);
    }
    // exclusive scans of the per-thread unions across the block, both directions (Hillis-Steele on inclusive values)
    ldsL[threadIdx.x] = mine; ldsR[threadIdx.x] = mine;
    __syncthreads();
    for (int s = 1; s < 256; s <<= 1) {
        Box l = ldsL[threadIdx.x], r = ldsR[threadIdx.x];
        if ((int)threadIdx.x >= s) l = box_union(ldsL[threadIdx.x - s], l);
        if ((int)threadIdx.x + s < 256) r = box_union(r, ldsR[threadIdx.x + s]);
        __syncthreads();
        ldsL[threadIdx.x] = l; ldsR[threadIdx.x] = r;
        __syncthreads();
    }
    Box left = box_union(tileLeft[blockIdx.x], threadIdx.x ? ldsL[threadIdx.x - 1] : box_identity());
    Box right = box_union(tileRight[blockIdx.x], threadIdx.x < 255 ? ldsR[threadIdx.x + 1] : box_identity());
    Box suf[kPerThread];
    Box acc = right;
    for (int k = kPerThread - 1; k >= 0; k--) { acc = box_union(acc, e[k]); suf[k] = acc; }
    float bestCost = 3.40282347e+38f;
    int bestIndex = -1;
    acc = left;
    for (int k = 0; k < kPerThread; k++) {
        const uint32_t i = first + k;
        if (i >= n) break;
        acc = box_union(acc, e[k]);
        const float parentSA = box_area(box_union(acc, suf[k]));
        const float cost = 1.f + ((box_area(acc) / parentSA) * (float)(int)i + (box_area(suf[k]) / parentSA) * (float)(n - i));
        if (cost < bestCost) { bestCost = cost; bestIndex = (int)i; }
    }
    // the tile's first minimum: lowest cost, lowest index among equals (threads hold ascending index ranges)
    ldsC[threadIdx.x] = bestCost; ldsI[threadIdx.x] = bestIndex;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const float c2 = ldsC[threadIdx.x + s]; const int i2 = ldsI[threadIdx.x + s];
            const float c1 = ldsC[threadIdx.x]; const int i1 = ldsI[threadIdx.x];
            if (i2 >= 0 && (i1 < 0 || c2 < c1 || (c2 == c1 && i2 < i1))) { ldsC[threadIdx.x] = c2; ldsI[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { tileCost[blockIdx.x] = ldsC[0]; tileIndex[blockIdx.x] = ldsI[0]; }
}

struct State {
    std::mutex mu;                 // host threads of the builder take turns on the device
    int device = -1;
    hipStream_t stream = nullptr;
    size_t cap = 0;                // triangles the buffers are sized for
    Box* dBoxes = nullptr; uint32_t* dIds = nullptr; Box* dSorted = nullptr;
    Box *dTileTotal = nullptr, *dTileLeft = nullptr, *dTileRight = nullptr;
    float* dTileCost = nullptr; int* dTileIndex = nullptr;
    std::vector<float> hCost; std::vector<int> hIndex;
    bool ok = false;
    bool busy = false;             // a build owns the device state from begin() to end(); a second build at that time sweeps on the host
};
inline State& state() { static State s; return s; }

inline void release_all(State& s) {
    void* p[] = {s.dBoxes, s.dIds, s.dSorted, s.dTileTotal, s.dTileLeft, s.dTileRight, s.dTileCost, s.dTileIndex};
    for (void* q : p) if (q) (void)hipFree(q);
    s.dBoxes = nullptr; s.dIds = nullptr; s.dSorted = nullptr; s.dTileTotal = s.dTileLeft = s.dTileRight = nullptr;
    s.dTileCost = nullptr; s.dTileIndex = nullptr;
    if (s.stream) { (void)hipStreamDestroy(s.stream); s.stream = nullptr; }
    s.cap = 0; s.ok = false;
}

// Makes the scene's triangle bounds resident on the device that is current in the calling (loader) thread.  false: no
// device / no memory / another build (another thread or context of this process) holds the state — the host sweeps then,
// with the same results.  One build at a time owns the state: begin() .. end() bracket it.
inline bool begin(const B6* boxes, size_t n) {
    State& s = state();
    std::lock_guard<std::mutex> g(s.mu);
    if (s.busy) return false;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return false; }
    if (hipGetDevice(&s.device) != hipSuccess) return false;
    release_all(s);
    const size_t tiles = (n + kTile - 1) / kTile;
    bool ok = hipStreamCreate(&s.stream) == hipSuccess;
    ok = ok && hipMalloc(&s.dBoxes, n * sizeof(Box)) == hipSuccess && hipMalloc(&s.dIds, n * 4) == hipSuccess && hipMalloc(&s.dSorted, n * sizeof(Box)) == hipSuccess;
    ok = ok && hipMalloc(&s.dTileTotal, tiles * sizeof(Box)) == hipSuccess && hipMalloc(&s.dTileLeft, tiles * sizeof(Box)) == hipSuccess &&
         hipMalloc(&s.dTileRight, tiles * sizeof(Box)) == hipSuccess && hipMalloc(&s.dTileCost, tiles * 4) == hipSuccess && hipMalloc(&s.dTileIndex, tiles * 4) == hipSuccess;
    ok = ok && hipMemcpy(s.dBoxes, boxes, n * sizeof(Box), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); release_all(s); return false; }
    s.hCost.resize(tiles); s.hIndex.resize(tiles);
    s.cap = n; s.ok = true; s.busy = true;
    return true;
}

inline bool run(const uint32_t* sortedIds, size_t n, float* cost, int* index) {
    State& s = state();
    std::lock_guard<std::mutex> g(s.mu);
    if (!s.ok || n > s.cap || n == 0) return false;
    if (hipSetDevice(s.device) != hipSuccess) return false;       // builder threads have no current device of their own
    const uint32_t tiles = (uint32_t)((n + kTile - 1) / kTile);
    bool ok = hipMemcpyAsync(s.dIds, sortedIds, n * 4, hipMemcpyHostToDevice, s.stream) == hipSuccess;
    if (ok) {
        k_sweep_gather<<<tiles, 256, 0, s.stream>>>(s.dBoxes, s.dIds, (uint32_t)n, s.dSorted, s.dTileTotal);
        k_sweep_tiles<<<1, 64, 0, s.stream>>>(s.dTileTotal, tiles, s.dTileLeft, s.dTileRight);
        k_sweep_cost<<<tiles, 256, 0, s.stream>>>(s.dSorted, (uint32_t)n, s.dTileLeft, s.dTileRight, s.dTileCost, s.dTileIndex);
        ok = hipGetLastError() == hipSuccess;
    }
    ok = ok && hipMemcpyAsync(s.hCost.data(), s.dTileCost, tiles * 4, hipMemcpyDeviceToHost, s.stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(s.hIndex.data(), s.dTileIndex, tiles * 4, hipMemcpyDeviceToHost, s.stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(s.stream) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); s.ok = false; return false; }     // the host sweeps from here on
    // model.cpp:445-458 over the tiles, in index order
    float bestCost = 3.40282347e+38f; int bestIndex = (int)(n / 2);
    for (uint32_t t = 0; t < tiles; t++) if (s.hIndex[t] >= 0 && s.hCost[t] < bestCost) { bestCost = s.hCost[t]; bestIndex = s.hIndex[t]; }
    *cost = bestCost; *index = bestIndex;
    return true;
}

inline void end() {
    State& s = state();
    std::lock_guard<std::mutex> g(s.mu);
    if (s.device >= 0) (void)hipSetDevice(s.device);
    release_all(s);
    s.busy = false;
}

struct Registrar {
    Registrar() {
        auto& G = rt_host::bvh_detail::gpu_sweep();
        G.begin = &begin; G.sweep = &run; G.end = &end;
    }
};
static Registrar g_registrar;

}  // namespace sweep
}  // namespace rt
