// Host SAH BVH builder producing the reference's slot-3 buffer bit for bit
// (replaces Model::computeForAxis / buildBVH / convertToGPU, /root/reference/Raytracing/model.cpp:417-530,
// and the BoundingBox / GPUBoundingBox constructors model.h:81-108,137-141).
//
// Same decisions as the reference — full sweep over 3 axes, libstdc++ std::sort with the `<` comparator on
// the centroid coordinate applied to the parent's winning order (tie order is part of the contract,
// SURVEY Q9), first strict minimum of the float32 cost, <= 4 triangles per leaf — but organised for
// throughput:
//  * structure-of-arrays keys and (key, index) pairs instead of pointers to 128-byte records.  A std::sort over
//    (key, id) pairs with a key-only comparator performs exactly the comparisons and moves of the reference's sort
//    over Triangle* (the algorithm never looks at anything but comparator results), so the permutation is identical;
//  * that std::sort itself runs on several threads without changing its result (stdSortPar below);
//  * an axis on which no two triangles of the whole scene share a centroid coordinate has exactly one sorted order per
//    node, whatever the sort and whatever the input order: such an axis is sorted ONCE for the scene and a node's order
//    is handed down by a stable split of its parent's (the classic O(n log n) SAH build).  Axes with ties — where
//    std::sort's own order is part of the tree — are still sorted per node, from the parent's winning order;
//  * the prefix / suffix sweeps read the node's bounds from one contiguous gather and run in chunks on several
//    threads (min / max are exact, so chunked scans give the sequential result), or on the GPU when the core has
//    registered its sweep kernels (large nodes only; same float operations, rt_core.hip);
//  * a bounded worker budget instead of three std::async threads per node.
#pragma once
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <stdexcept>
#include <thread>
#include <vector>
#include <sys/resource.h>
#include "../rt_types.h"

namespace rt_host {

using rt::RefNode;
using rt::RefTriangle;

namespace bvh_detail {

struct B3 { float x, y, z; };
inline B3 bmin(B3 a, B3 b) { return {b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z}; }   // glm::min(x, y) = y < x ? y : x
inline B3 bmax(B3 a, B3 b) { return {a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z}; }   // glm::max(x, y) = x < y ? y : x
inline float area(B3 mn, B3 mx) {                                                                                  // model.h:115-118
    float ex = mx.x - mn.x, ey = mx.y - mn.y, ez = mx.z - mn.z;
    return 2.f * (ex * ey + ey * ez + ez * ex);
}

struct KeyId { float key; uint32_t id; };
struct B6 { B3 lo, hi; };            // Triangle::minBounds / maxBounds

struct Soa {
    std::vector<float> cen[3];
    std::vector<B6> box;
    const RefTriangle* tris = nullptr;
};

struct Node {                        // BoundingBox, model.h:72-108
    B3 lo, hi;
    std::unique_ptr<Node> left, right;
    uint32_t tri[4]; uint32_t ntri = 0; bool leaf = false;
};

struct AxisResult { float cost; int index; };

// Optional device side of the sweep (registered by rt_core.hip): computeForAxis's prefix / suffix boxes, cost and first
// strict minimum for a node's triangles in sorted order, on the GPU.  begin() makes the scene's bounds resident and says
// whether the device can be used; sweep() returns false when it cannot run (then the host sweep does).
struct GpuSweep {
    bool (*begin)(const B6* boxes, size_t n) = nullptr;
    bool (*sweep)(const uint32_t* sortedIds, size_t n, float* cost, int* index) = nullptr;
    void (*end)() = nullptr;
    size_t minNode = (size_t)1 << 18;                                         // smaller nodes stay on the host
};
inline GpuSweep& gpu_sweep() { static GpuSweep g; return g; }

// RT_BVH_TIMING=1: thread-time spent per phase (nanoseconds, summed over threads) and the wall time of the levels
struct BuildTimes { std::atomic<long long> sortNs{0}, gatherNs{0}, sweepNs{0}, gpuNs{0}, splitNs{0}, leafNs{0}, waitNs{0}; bool on = false; };
inline BuildTimes& build_times() { static BuildTimes t; return t; }
struct PhaseTimer {
    std::atomic<long long>* acc; std::chrono::steady_clock::time_point t0;
    explicit PhaseTimer(std::atomic<long long>& a) : acc(build_times().on ? &a : nullptr) { if (acc) t0 = std::chrono::steady_clock::now(); }
    ~PhaseTimer() { if (acc) acc->fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count()); }
};

// Per-thread scratch of the sorts and sweeps.  A node's temporaries used to be fresh vectors: above 128 KB each of them is
// an mmap / page-fault / munmap round trip, and with 16 threads doing that for every node the kernel's address-space lock
// became the builder's bottleneck (thread time 44 s, wall 5.6 s at 20 M triangles).  Grow-only, released at the end of
// build_bvh on the calling thread and at thread exit on the workers.
struct Scratch { std::vector<KeyId> sorted; std::vector<B6> b, suf; };
inline Scratch& scratch() { static thread_local Scratch s; return s; }

struct Builder {
    const Soa& S;
    std::atomic<int> freeWorkers;
    bool distinct[3] = {false, false, false};     // no two triangles of the scene share a centroid coordinate on this axis
    std::vector<uint8_t> side;                    // scratch of the stable splits: which child a triangle went to
    bool useGpu = false;
    explicit Builder(const Soa& s, int workers) : S(s), freeWorkers(workers) {}

    bool tryAcquire(int k) {
        int cur = freeWorkers.load();
        while (cur >= k) if (freeWorkers.compare_exchange_weak(cur, cur - k)) return true;
        return false;
    }
    void release(int k) { freeWorkers.fetch_add(k); }
    // A thread that blocks on its helpers lends its own slot out for that time (the budget may dip below zero for a moment
    // when it resumes; tryAcquire then fails until the balance is back).  Without this up to half of the threads sat in
    // future::get() holding a slot: 3.9 of 8 cores busy on a 2 M-triangle build.
    struct Lend {
        Builder& b;
        explicit Lend(Builder& bb) : b(bb) { b.freeWorkers.fetch_add(1); }
        ~Lend() { b.freeWorkers.fetch_sub(1); }
    };

    // f(chunkIndex, begin, end) over [0, n) in `chunks` pieces, on as many threads as the budget allows
    template <typename F>
    void parallelChunks(size_t n, size_t chunks, F f) {
        if (chunks < 1) chunks = 1;
        const size_t per = (n + chunks - 1) / chunks;
        std::vector<std::future<void>> kids;
        for (size_t c = 0; c < chunks; c++) {
            const size_t b = c * per, e = std::min(n, b + per);
            if (b >= e) break;
            if (e < n && tryAcquire(1)) kids.push_back(std::async(std::launch::async, [this, &f, c, b, e] { f(c, b, e); release(1); }));
            else f(c, b, e);
        }
        if (!kids.empty()) { Lend lend(*this); for (auto& k : kids) k.get(); }
    }

    // std::sort(first, last, comp) — the very permutation, tie order included — with the work spread over threads.
    // libstdc++'s sort is __introsort_loop (median-of-three to the front, unguarded partition, recursion on the upper part,
    // loop on the lower, heapsort when the depth budget runs out) followed by one insertion sort over the whole range.
    // The two sides of a partition never touch each other's elements, so the recursive call may as well run on another
    // thread: the same library routines make the same comparisons and moves on the same sub-ranges in any order.  The final
    // insertion sort is stable and only ever moves an element within the <= 16-element range the loop left it in, so it
    // equals per-chunk insertion sorts plus a 32-element insertion sort across every chunk seam.
    static constexpr ptrdiff_t kSortTask = 1 << 16;      // sub-ranges below this are sorted by the calling thread
    template <typename It, typename Cmp>
    void introsortPar(It first, It last, long depth, Cmp comp) {
        std::vector<std::future<void>> kids;
        while (last - first > 16) {
            if (last - first < kSortTask) { std::__introsort_loop(first, last, depth, comp); break; }
            if (depth == 0) { std::__partial_sort(first, last, last, comp); break; }
            --depth;
            It cut = std::__unguarded_partition_pivot(first, last, comp);
            if (last - cut >= kSortTask && tryAcquire(1))
                kids.push_back(std::async(std::launch::async, [this, cut, last, depth, comp] { introsortPar(cut, last, depth, comp); release(1); }));
            else introsortPar(cut, last, depth, comp);
            last = cut;
        }
        if (!kids.empty()) { Lend lend(*this); for (auto& k : kids) k.get(); }
    }
    template <typename It, typename Less>
    void stdSortPar(It first, It last, Less less) {
        const ptrdiff_t n = last - first;
        if (n < 4 * kSortTask || serialSweeps || std::getenv("RT_BVH_SEQ_SORT")) { std::sort(first, last, less); return; }
        auto comp = __gnu_cxx::__ops::__iter_comp_iter(less);
        introsortPar(first, last, (long)std::__lg(n) * 2, comp);
        const size_t chunks = (size_t)std::max<ptrdiff_t>(1, std::min<ptrdiff_t>(64, n / kSortTask));
        const ptrdiff_t per = (ptrdiff_t)(((size_t)n + chunks - 1) / chunks);      // parallelChunks' own chunk length
        parallelChunks((size_t)n, chunks, [&](size_t, size_t b, size_t e) { std::__insertion_sort(first + b, first + e, comp); });
        for (ptrdiff_t p = per; p < n; p += per) std::__insertion_sort(first + std::max<ptrdiff_t>(0, p - 16), first + std::min(n, p + 16), comp);
    }

    // model.cpp:425-461 on the node's triangles in sorted order (ids[i] = i-th triangle of the sorted copy).
    // (Chunked scans: min and max of non-NaN floats are exact, associative and commutative up to the sign of a zero, and
    // these boxes only feed differences — area() — so the chunked sweep returns the sequential (cost, index).  Scenes with
    // a NaN bound take the single-chunk path, see build_bvh.)
    AxisResult sweepSorted(const uint32_t* ids, size_t n) {
        if (useGpu && n >= gpu_sweep().minNode) {
            PhaseTimer pt(build_times().gpuNs);
            AxisResult r;
            if (gpu_sweep().sweep(ids, n, &r.cost, &r.index)) return r;
        }
        PhaseTimer pt(build_times().sweepNs);
        Scratch& scr = scratch();
        if (scr.b.size() < n) { scr.b.resize(n); scr.suf.resize(n); }
        B6* const b = scr.b.data();
        B6* const suf = scr.suf.data();
        const size_t chunks = (n >= ((size_t)1 << 17) && freeWorkers.load() >= 0 && !serialSweeps) ? std::min<size_t>(64, n >> 15) : 1;
        const size_t per = (n + chunks - 1) / chunks;
        std::vector<B6> total(chunks);
        // pass 1: contiguous copy of the bounds in sorted order + each chunk's union
        parallelChunks(n, chunks, [&](size_t c, size_t lo, size_t hi) {
            B6 t = S.box[ids[lo]];
            for (size_t i = lo; i < hi; i++) { b[i] = S.box[ids[i]]; t.lo = bmin(t.lo, b[i].lo); t.hi = bmax(t.hi, b[i].hi); }
            total[c] = t;
        });
        // pass 2: inclusive suffix and prefix boxes (chunk-local scans seeded with the neighbours' unions), cost, first minimum
        std::vector<AxisResult> best(chunks, AxisResult{FLT_MAX, -1});
        parallelChunks(n, chunks, [&](size_t c, size_t lo, size_t hi) {
            bool haveR = false, haveL = false;
            B6 right{}, left{};
            for (size_t k = c + 1; k < chunks && k * per < n; k++) { if (!haveR) { right = total[k]; haveR = true; } else { right.lo = bmin(right.lo, total[k].lo); right.hi = bmax(right.hi, total[k].hi); } }
            for (size_t k = 0; k < c; k++) { if (!haveL) { left = total[k]; haveL = true; } else { left.lo = bmin(left.lo, total[k].lo); left.hi = bmax(left.hi, total[k].hi); } }
            for (size_t i = hi; i-- > lo;) {
                if (i == hi - 1) suf[i] = haveR ? B6{bmin(right.lo, b[i].lo), bmax(right.hi, b[i].hi)} : b[i];
                else suf[i] = B6{bmin(suf[i + 1].lo, b[i].lo), bmax(suf[i + 1].hi, b[i].hi)};
            }
            B6 pre{};
            float bestCost = FLT_MAX; int bestIndex = -1;
            for (size_t i = lo; i < hi; i++) {
                if (i == lo) pre = haveL ? B6{bmin(left.lo, b[i].lo), bmax(left.hi, b[i].hi)} : b[i];
                else pre = B6{bmin(pre.lo, b[i].lo), bmax(pre.hi, b[i].hi)};
                float parentSA = area(bmin(pre.lo, suf[i].lo), bmax(pre.hi, suf[i].hi));
                float cost = 1.f + ((area(pre.lo, pre.hi) / parentSA) * (float)(int)i + (area(suf[i].lo, suf[i].hi) / parentSA) * (float)(n - i));
                if (cost < bestCost) { bestCost = cost; bestIndex = (int)i; }
            }
            best[c] = {bestCost, bestIndex};
        });
        // model.cpp:445-458: the first index whose cost is below every earlier one, starting from FLT_MAX
        AxisResult r{FLT_MAX, (int)(n / 2)};
        for (size_t c = 0; c < chunks; c++) if (best[c].index >= 0 && best[c].cost < r.cost) r = best[c];
        return r;
    }
    bool serialSweeps = false;

    std::unique_ptr<Node> makeLeaf(const uint32_t* ids, size_t n) const {   // model.h:94-108
        PhaseTimer pt(build_times().leafNs);
        auto nd = std::make_unique<Node>();
        nd->leaf = true; nd->ntri = (uint32_t)n;
        nd->lo = {FLT_MAX, FLT_MAX, FLT_MAX}; nd->hi = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (size_t i = 0; i < n; i++) {
            nd->tri[i] = ids[i];
            const RefTriangle& t = S.tris[ids[i]];
            const rt::f3 pts[3] = {t.p1, t.p2, t.p3};
            for (const rt::f3& p : pts) { B3 q{p.x, p.y, p.z}; nd->lo = bmin(nd->lo, q); nd->hi = bmax(nd->hi, q); }
        }
        return nd;
    }

    // model.cpp:467-513.  ids: the node's triangles in the parent's winning order (the input of the reference's sorts).
    // pre[a]: for an axis with distinct keys, the node's triangles in that axis' (unique) sorted order; empty otherwise.
    std::unique_ptr<Node> build(std::vector<uint32_t> ids, std::vector<uint32_t> pre[3], int depth) {
        const size_t n = ids.size();
        if (n <= 4) return makeLeaf(ids.data(), n);
        if (depth > 4096) throw std::runtime_error("BVH build: degenerate input (the reference recurses without bound here)");
        std::vector<uint32_t> order[3];
        AxisResult r[3];
        const bool timed = build_times().on && depth <= 4;
        const auto tn0 = std::chrono::steady_clock::now();
        auto doAxis = [&](int a) {
            if (distinct[a]) order[a] = std::move(pre[a]);
            else {
                PhaseTimer pt(build_times().sortNs);
                std::vector<KeyId>& sorted = scratch().sorted;
                if (sorted.size() < n) sorted.resize(n);
                const float* cen = S.cen[a].data();
                // (the gathers of the few huge nodes at the top run in chunks on the free workers, like their sorts)
                const size_t gchunks = (n >= ((size_t)1 << 20) && !serialSweeps) ? 16 : 1;
                const uint32_t* idp = ids.data();
                KeyId* sp = sorted.data();
                parallelChunks(n, gchunks, [&](size_t, size_t b, size_t e) { for (size_t i = b; i < e; i++) { sp[i].key = cen[idp[i]]; sp[i].id = idp[i]; } });
                stdSortPar(sorted.begin(), sorted.begin() + (ptrdiff_t)n, [](const KeyId& x, const KeyId& y) { return x.key < y.key; });
                order[a].resize(n);
                uint32_t* op = order[a].data();
                parallelChunks(n, gchunks, [&](size_t, size_t b, size_t e) { for (size_t i = b; i < e; i++) op[i] = sp[i].id; });
            }
            r[a] = sweepSorted(order[a].data(), n);
        };
        if (n >= 16384 && tryAcquire(2)) {
            auto f0 = std::async(std::launch::async, [&] { doAxis(0); });
            auto f1 = std::async(std::launch::async, [&] { doAxis(1); });
            try { doAxis(2); } catch (...) { f0.wait(); f1.wait(); release(2); throw; }
            { PhaseTimer pw(build_times().waitNs); Lend lend(*this); f0.get(); f1.get(); }
            release(2);
        } else {
            for (int a = 0; a < 3; a++) doAxis(a);
        }
        const auto tn1 = std::chrono::steady_clock::now();
        float best = r[0].cost; int axis = 0, split = r[0].index;
        if (r[1].cost < best) { best = r[1].cost; axis = 1; split = r[1].index; }
        if (r[2].cost < best) { best = r[2].cost; axis = 2; split = r[2].index; }
        if (split == 0) throw std::runtime_error("BVH build: SAH split at index 0 (identical bounds); the reference recurses without bound on this input");
        std::unique_ptr<PhaseTimer> splitTimer(new PhaseTimer(build_times().splitNs));
        std::vector<uint32_t> ids0(order[axis].begin(), order[axis].begin() + split), ids1(order[axis].begin() + split, order[axis].end());
        // the children's orders on the axes with distinct keys: the winning axis' is the two halves themselves, another
        // axis' is a stable split of this node's order by the side each triangle went to
        std::vector<uint32_t> pre0[3], pre1[3];
        bool marked = false;
        for (int a = 0; a < 3; a++) {
            if (!distinct[a]) continue;
            if (a == axis) { pre0[a] = ids0; pre1[a] = ids1; continue; }
            if (!marked) {
                for (uint32_t id : ids0) side[id] = 0;
                for (uint32_t id : ids1) side[id] = 1;
                marked = true;
            }
            pre0[a].reserve(ids0.size()); pre1[a].reserve(ids1.size());
            for (uint32_t id : order[a]) (side[id] ? pre1[a] : pre0[a]).push_back(id);
        }
        for (int a = 0; a < 3; a++) std::vector<uint32_t>().swap(order[a]);
        std::vector<uint32_t>().swap(ids);
        splitTimer.reset();
        if (timed) {
            const auto tn2 = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[bvh] depth %d, %zu triangles: sorts + sweeps %.3f s, split %.3f s (axis %d at %d)\n", depth, n,
                         std::chrono::duration<double>(tn1 - tn0).count(), std::chrono::duration<double>(tn2 - tn1).count(), axis, split);
        }
        std::unique_ptr<Node> left, right;
        if (n >= 4096 && tryAcquire(1)) {
            auto lf = std::async(std::launch::async, [&] { return build(std::move(ids0), pre0, depth + 1); });
            try { right = build(std::move(ids1), pre1, depth + 1); } catch (...) { lf.wait(); release(1); throw; }
            { PhaseTimer pw(build_times().waitNs); Lend lend(*this); left = lf.get(); }
            release(1);
        } else {
            left = build(std::move(ids0), pre0, depth + 1);
            right = build(std::move(ids1), pre1, depth + 1);
        }
        auto nd = std::make_unique<Node>();                                   // model.h:81-92
        nd->lo = bmin(left->lo, right->lo);
        nd->hi = bmax(left->hi, right->hi);
        nd->left = std::move(left); nd->right = std::move(right);
        return nd;
    }
};

// model.cpp:515-530 + model.h:137-141, iterative (pre-order, left subtree before right)
inline void flatten(const Node* root, std::vector<RefNode>& out) {
    struct Item { const Node* n; int32_t parent; int which; };
    std::vector<Item> st;
    st.push_back({root, -1, 0});
    while (!st.empty()) {
        Item it = st.back(); st.pop_back();
        uint32_t cur = (uint32_t)out.size();
        RefNode g; std::memset(&g, 0, sizeof g);
        const Node* n = it.n;
        if (n->leaf) {
            g.minLoc = {FLT_MAX, FLT_MAX, FLT_MAX}; g.maxLoc = {-FLT_MAX, -FLT_MAX, -FLT_MAX};   // no children: both ?: take the sentinel
            g.isLeaf = 1;
            for (int k = 0; k < 4; k++) g.children[k] = k < (int)n->ntri ? (int32_t)n->tri[k] : -1;
        } else {
            B3 lo = bmin(n->left->lo, n->right->lo), hi = bmax(n->left->hi, n->right->hi);
            g.minLoc = {lo.x, lo.y, lo.z}; g.maxLoc = {hi.x, hi.y, hi.z};
            g.isLeaf = 0;
            g.children[0] = g.children[1] = g.children[2] = g.children[3] = -1;
        }
        out.push_back(g);
        if (it.parent >= 0) out[it.parent].children[it.which] = (int32_t)cur;
        if (!n->leaf) {
            st.push_back({n->right.get(), (int32_t)cur, 1});   // popped second
            st.push_back({n->left.get(), (int32_t)cur, 0});    // popped first -> left subtree numbered first
        }
    }
}

inline int default_workers() {
    int w = (int)std::thread::hardware_concurrency();
    // a job's CPU share, when the environment states one (the GPU boxes do: 16 per GPU), bounds the worker pool
    if (const char* e = std::getenv("OMP_NUM_THREADS")) { int v = std::atoi(e); if (v > 0 && v < w) w = v; }
    if (w > 32) w = 32;
    if (w < 1) w = 1;
    return w;
}

}  // namespace bvh_detail

// Builds the DFS pre-order GPUBoundingBox list for `n` Triangle records (original order).
// allowGpu: the prefix / suffix sweeps of large nodes may run on the device, if the core has registered them.
inline std::vector<RefNode> build_bvh(const RefTriangle* tris, size_t n, int workers = 0, bool allowGpu = false) {
    using namespace bvh_detail;
    Soa S;
    S.tris = tris;
    for (int a = 0; a < 3; a++) S.cen[a].resize(n);
    S.box.resize(n);
    if (workers <= 0) workers = default_workers();
    BuildTimes& BT = build_times();
    BT.on = std::getenv("RT_BVH_TIMING") != nullptr;
    BT.sortNs = 0; BT.gatherNs = 0; BT.sweepNs = 0; BT.gpuNs = 0; BT.splitNs = 0; BT.leafNs = 0; BT.waitNs = 0;
    const auto wall0 = std::chrono::steady_clock::now();
    auto cpuNow = [] { rusage u; getrusage(RUSAGE_SELF, &u); return (double)u.ru_utime.tv_sec + u.ru_utime.tv_usec * 1e-6 + (double)u.ru_stime.tv_sec + u.ru_stime.tv_usec * 1e-6; };
    auto sysNow = [] { rusage u; getrusage(RUSAGE_SELF, &u); return (double)u.ru_stime.tv_sec + u.ru_stime.tv_usec * 1e-6; };
    const double cpu0 = cpuNow(), sys0 = sysNow();
    auto wall = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count(); };
    double tSetup = 0, tProbe = 0, tBuild = 0, tFlat = 0;
    Builder B(S, workers - 1);
    std::atomic<bool> nan{false};
    B.parallelChunks(n, n >= ((size_t)1 << 18) ? 32 : 1, [&](size_t, size_t b, size_t e) {
        bool bad = false;
        for (size_t i = b; i < e; i++) {
            S.cen[0][i] = tris[i].centroidLoc.x; S.cen[1][i] = tris[i].centroidLoc.y; S.cen[2][i] = tris[i].centroidLoc.z;
            S.box[i].lo = {tris[i].minBounds.x, tris[i].minBounds.y, tris[i].minBounds.z};
            S.box[i].hi = {tris[i].maxBounds.x, tris[i].maxBounds.y, tris[i].maxBounds.z};
            const B6& q = S.box[i];
            if (q.lo.x != q.lo.x || q.lo.y != q.lo.y || q.lo.z != q.lo.z || q.hi.x != q.hi.x || q.hi.y != q.hi.y || q.hi.z != q.hi.z) bad = true;
            for (int a = 0; a < 3; a++) if (S.cen[a][i] != S.cen[a][i]) bad = true;
        }
        if (bad) nan = true;
    });
    std::vector<uint32_t> ids(n);
    for (size_t i = 0; i < n; i++) ids[i] = (uint32_t)i;
    std::vector<uint32_t> pre[3];
    tSetup = wall();
    // With a NaN anywhere the `?:` min / max and the comparator stop being order-independent: such a scene is built
    // exactly as the reference writes it (sequential sweeps, a sort per node and axis, no device sweeps).
    if (nan) B.serialSweeps = true;
    else if (n > 4 && !std::getenv("RT_BVH_NO_PRESORT")) {
        // which axes have pairwise distinct keys: one sort of the whole scene per axis (any correct sort will do here)
        auto probe = [&](int a) {
            const float* cen = S.cen[a].data();
            // a strided sample first: a grid-aligned scene (config 4's field on x and z) shows its ties in a few thousand
            // keys, and the scene-wide sort is then not worth making
            if (n > ((size_t)1 << 20)) {
                const size_t step = n >> 18;
                std::vector<float> smp;
                smp.reserve(n / step + 1);
                for (size_t i = 0; i < n; i += step) smp.push_back(cen[i]);
                std::sort(smp.begin(), smp.end());
                for (size_t i = 1; i < smp.size(); i++) if (!(smp[i - 1] < smp[i])) { B.distinct[a] = false; return; }
            }
            std::vector<KeyId> k(n);
            for (size_t i = 0; i < n; i++) { k[i].key = cen[i]; k[i].id = (uint32_t)i; }
            B.stdSortPar(k.begin(), k.end(), [](const KeyId& x, const KeyId& y) { return x.key < y.key; });
            bool dist = true;
            for (size_t i = 1; i < n && dist; i++) if (!(k[i - 1].key < k[i].key)) dist = false;
            if (dist) { pre[a].resize(n); for (size_t i = 0; i < n; i++) pre[a][i] = k[i].id; }
            B.distinct[a] = dist;
        };
        if (n >= 16384 && B.tryAcquire(2)) {
            auto f0 = std::async(std::launch::async, [&] { probe(0); });
            auto f1 = std::async(std::launch::async, [&] { probe(1); });
            probe(2);
            f0.get(); f1.get();
            B.release(2);
        } else for (int a = 0; a < 3; a++) probe(a);
        if (B.distinct[0] || B.distinct[1] || B.distinct[2]) B.side.assign(n, 0);
    }
    tProbe = wall();
    GpuSweep& G = gpu_sweep();
    B.useGpu = allowGpu && !nan && G.begin && G.sweep && n >= G.minNode && !std::getenv("RT_BVH_NO_GPU") && G.begin(S.box.data(), n);
    std::unique_ptr<Node> root;
    try { root = B.build(std::move(ids), pre, 0); }
    catch (...) { if (B.useGpu && G.end) G.end(); throw; }
    if (B.useGpu && G.end) G.end();
    { Scratch& scr = scratch(); std::vector<KeyId>().swap(scr.sorted); std::vector<B6>().swap(scr.b); std::vector<B6>().swap(scr.suf); }
    tBuild = wall();
    std::vector<RefNode> out;
    out.reserve(n);
    flatten(root.get(), out);
    tFlat = wall();
    if (BT.on)
        std::fprintf(stderr, "[bvh] %zu triangles, %d workers, gpu sweeps %d: wall setup %.3f probe %.3f build %.3f flatten %.3f s; thread time sort %.3f host sweep %.3f gpu sweep %.3f split %.3f leaves %.3f waiting for a sibling %.3f s; process CPU time %.3f s (system %.3f)\n",
                     n, workers, (int)B.useGpu, tSetup, tProbe - tSetup, tBuild - tProbe, tFlat - tBuild, BT.sortNs.load() * 1e-9, BT.sweepNs.load() * 1e-9, BT.gpuNs.load() * 1e-9, BT.splitNs.load() * 1e-9, BT.leafNs.load() * 1e-9, BT.waitNs.load() * 1e-9, cpuNow() - cpu0, sysNow() - sys0);
    // deep trees: release iteratively to keep the destructor recursion shallow
    std::vector<std::unique_ptr<Node>> pend;
    pend.push_back(std::move(root));
    while (!pend.empty()) {
        std::unique_ptr<Node> x = std::move(pend.back()); pend.pop_back();
        if (x->left) pend.push_back(std::move(x->left));
        if (x->right) pend.push_back(std::move(x->right));
    }
    return out;
}

}  // namespace rt_host
