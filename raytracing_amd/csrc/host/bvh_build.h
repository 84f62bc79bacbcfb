// Host SAH BVH builder producing the reference's slot-3 buffer bit for bit
// (replaces Model::computeForAxis / buildBVH / convertToGPU, /root/reference/Raytracing/model.cpp:417-530,
// and the BoundingBox / GPUBoundingBox constructors model.h:81-108,137-141).
//
// Same decisions as the reference — full sweep over 3 axes, libstdc++ std::sort with the `<` comparator on
// the centroid coordinate applied to the parent's winning order (tie order is part of the contract,
// SURVEY Q9), first strict minimum of the float32 cost, <= 4 triangles per leaf — but organised for
// throughput: structure-of-arrays keys, (key, index) pairs instead of pointers to 128-byte records,
// per-node scratch reuse, and a bounded task pool instead of three std::async threads per node.
// A std::sort over (key, id) pairs with a key-only comparator performs exactly the comparisons and moves
// of the reference's sort over Triangle* (the algorithm never looks at anything but comparator results),
// so the permutation — and therefore the tree — is identical.
#pragma once
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cstdint>
#include <cstring>
#include <future>
#include <memory>
#include <stdexcept>
#include <vector>
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

struct Soa {
    std::vector<float> cen[3];
    std::vector<B3> lo, hi;          // Triangle::minBounds / maxBounds
    const RefTriangle* tris = nullptr;
};

struct Node {                        // BoundingBox, model.h:72-108
    B3 lo, hi;
    std::unique_ptr<Node> left, right;
    uint32_t tri[4]; uint32_t ntri = 0; bool leaf = false;
};

struct Builder {
    const Soa& S;
    std::atomic<int> freeWorkers;
    explicit Builder(const Soa& s, int workers) : S(s), freeWorkers(workers) {}

    struct AxisResult { float cost; int index; };

    // model.cpp:417-462
    AxisResult sweepAxis(const uint32_t* ids, size_t n, int axis, std::vector<KeyId>& sorted,
                         std::vector<B3>& sufLo, std::vector<B3>& sufHi) const {
        sorted.resize(n);
        const float* cen = S.cen[axis].data();
        for (size_t i = 0; i < n; i++) { sorted[i].key = cen[ids[i]]; sorted[i].id = ids[i]; }
        std::sort(sorted.begin(), sorted.end(), [](const KeyId& a, const KeyId& b) { return a.key < b.key; });
        sufLo.resize(n); sufHi.resize(n);
        sufLo[n - 1] = S.lo[sorted[n - 1].id]; sufHi[n - 1] = S.hi[sorted[n - 1].id];
        for (size_t i = n - 1; i-- > 0;) {
            sufLo[i] = bmin(sufLo[i + 1], S.lo[sorted[i].id]);
            sufHi[i] = bmax(sufHi[i + 1], S.hi[sorted[i].id]);
        }
        float bestCost = FLT_MAX;
        int bestIndex = (int)(n / 2);
        B3 preLo = S.lo[sorted[0].id], preHi = S.hi[sorted[0].id];
        for (size_t i = 0; i < n; i++) {
            if (i) { preLo = bmin(preLo, S.lo[sorted[i].id]); preHi = bmax(preHi, S.hi[sorted[i].id]); }
            float parentSA = area(bmin(preLo, sufLo[i]), bmax(preHi, sufHi[i]));
            float cost = 1.f + ((area(preLo, preHi) / parentSA) * (float)(int)i + (area(sufLo[i], sufHi[i]) / parentSA) * (float)(n - i));
            if (cost < bestCost) { bestCost = cost; bestIndex = (int)i; }
        }
        return {bestCost, bestIndex};
    }

    std::unique_ptr<Node> makeLeaf(const uint32_t* ids, size_t n) const {   // model.h:94-108
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

    // model.cpp:467-513
    std::unique_ptr<Node> build(const uint32_t* ids, size_t n, int depth) {
        if (n <= 4) return makeLeaf(ids, n);
        if (depth > 4096) throw std::runtime_error("BVH build: degenerate input (the reference recurses without bound here)");
        std::vector<KeyId> sorted[3];
        std::vector<B3> sufLo[3], sufHi[3];
        AxisResult r[3];
        bool par = n >= 16384 && tryAcquire(2);
        if (par) {
            auto f0 = std::async(std::launch::async, [&] { return sweepAxis(ids, n, 0, sorted[0], sufLo[0], sufHi[0]); });
            auto f1 = std::async(std::launch::async, [&] { return sweepAxis(ids, n, 1, sorted[1], sufLo[1], sufHi[1]); });
            r[2] = sweepAxis(ids, n, 2, sorted[2], sufLo[2], sufHi[2]);
            r[0] = f0.get(); r[1] = f1.get();
            release(2);
        } else {
            for (int a = 0; a < 3; a++) r[a] = sweepAxis(ids, n, a, sorted[a], sufLo[a], sufHi[a]);
        }
        float best = r[0].cost; int axis = 0, split = r[0].index;
        if (r[1].cost < best) { best = r[1].cost; axis = 1; split = r[1].index; }
        if (r[2].cost < best) { best = r[2].cost; axis = 2; split = r[2].index; }
        std::vector<uint32_t> order(n);
        for (size_t i = 0; i < n; i++) order[i] = sorted[axis][i].id;
        for (int a = 0; a < 3; a++) { std::vector<KeyId>().swap(sorted[a]); std::vector<B3>().swap(sufLo[a]); std::vector<B3>().swap(sufHi[a]); }
        if (split == 0) throw std::runtime_error("BVH build: SAH split at index 0 (identical bounds); the reference recurses without bound on this input");
        std::unique_ptr<Node> left, right;
        if (n >= 4096 && tryAcquire(1)) {
            auto lf = std::async(std::launch::async, [&] { return build(order.data(), (size_t)split, depth + 1); });
            try { right = build(order.data() + split, n - (size_t)split, depth + 1); } catch (...) { lf.wait(); release(1); throw; }
            left = lf.get();
            release(1);
        } else {
            left = build(order.data(), (size_t)split, depth + 1);
            right = build(order.data() + split, n - (size_t)split, depth + 1);
        }
        auto nd = std::make_unique<Node>();                                   // model.h:81-92
        nd->lo = bmin(left->lo, right->lo);
        nd->hi = bmax(left->hi, right->hi);
        nd->left = std::move(left); nd->right = std::move(right);
        return nd;
    }

    bool tryAcquire(int k) {
        int cur = freeWorkers.load();
        while (cur >= k) if (freeWorkers.compare_exchange_weak(cur, cur - k)) return true;
        return false;
    }
    void release(int k) { freeWorkers.fetch_add(k); }
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

}  // namespace bvh_detail

// Builds the DFS pre-order GPUBoundingBox list for `n` Triangle records (original order).
inline std::vector<RefNode> build_bvh(const RefTriangle* tris, size_t n, int workers = 0) {
    using namespace bvh_detail;
    Soa S;
    S.tris = tris;
    for (int a = 0; a < 3; a++) S.cen[a].resize(n);
    S.lo.resize(n); S.hi.resize(n);
    std::vector<uint32_t> ids(n);
    for (size_t i = 0; i < n; i++) {
        S.cen[0][i] = tris[i].centroidLoc.x; S.cen[1][i] = tris[i].centroidLoc.y; S.cen[2][i] = tris[i].centroidLoc.z;
        S.lo[i] = {tris[i].minBounds.x, tris[i].minBounds.y, tris[i].minBounds.z};
        S.hi[i] = {tris[i].maxBounds.x, tris[i].maxBounds.y, tris[i].maxBounds.z};
        ids[i] = (uint32_t)i;
    }
    if (workers <= 0) { workers = (int)std::thread::hardware_concurrency(); if (workers > 32) workers = 32; if (workers < 1) workers = 1; }
    Builder B(S, workers - 1);
    std::unique_ptr<Node> root = B.build(ids.data(), n, 0);
    std::vector<RefNode> out;
    out.reserve(n);
    flatten(root.get(), out);
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
