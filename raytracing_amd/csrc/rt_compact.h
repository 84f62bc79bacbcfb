// Host-side derivation of the compacted device layout (CNode / CTri / CShade) from the reference's
// SSBO images (slots 0,1,3,4).  Runs once per scene upload, on the host, with the same float
// operations the shader performs per ray (shader.comp:263-265, 348-350), so the kernels read values
// bit-identical to what the reference recomputes on every test.  Compiled with -ffp-contract=off.
#pragma once
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include "rt_types.h"

namespace rt {

// An array that is not value-initialised when it is sized: the big per-triangle tables are written in full by several
// threads, and zero-filling gigabytes on one thread first (std::vector::resize) took longer than filling them.
template <typename T>
struct RawBuf {
    std::unique_ptr<T[]> p;
    size_t n = 0;
    void alloc(size_t k) { p.reset(k ? new T[k] : nullptr); n = k; }
    void clear() { p.reset(); n = 0; }
    T* data() { return p.get(); }
    const T* data() const { return p.get(); }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

// f(lo, hi, chunk) over [0, n) on up to 32 threads (the job's CPU share when OMP_NUM_THREADS states one); one thread below `serialBelow`
template <typename F>
static inline unsigned parallel_ranges(uint64_t n, uint64_t serialBelow, F f) {
    unsigned nThreads = 1;
    if (n >= serialBelow) {
        nThreads = std::thread::hardware_concurrency();
        if (const char* e = std::getenv("OMP_NUM_THREADS")) { int v = std::atoi(e); if (v > 0 && (unsigned)v < nThreads) nThreads = (unsigned)v; }
        nThreads = std::max(1u, std::min(nThreads, 32u));
    }
    const uint64_t per = (n + nThreads - 1) / nThreads;
    std::vector<std::thread> th;
    for (unsigned k = 1; k < nThreads; k++) th.emplace_back(f, std::min<uint64_t>(n, k * per), std::min<uint64_t>(n, (k + 1) * per), k);
    f(0, std::min<uint64_t>(n, per), 0u);
    for (auto& t : th) t.join();
    return nThreads;
}

struct CompactScene {
    RawBuf<CNode> nodes;
    std::vector<WNode> wnodes;      // two-level view of `nodes` (empty when the tree does not allow it, see derive_wide)
    RawBuf<CTri> tris;              // (released after the upload: nTris keeps the count)
    RawBuf<CShade> shade;
    RawBuf<CShadeUV> shadeUV;       // filled by fill_shade_uv, only for scenes with a textured material
    uint64_t nTris = 0;
    int32_t rootChild = -1;
    float rootLo[3] = {0, 0, 0}, rootHi[3] = {0, 0, 0};
    float sceneScale = 1.f;
    bool boxesContainTris = true;   // false -> only traversalMode 1 is valid (mesh-local boxes vs world triangles)
    uint32_t maxDepth = 0;
    uint32_t wideDepth = 0;         // levels of wide nodes on the longest root-to-leaf path
};

// vec3(M * vec4(v, w)), column-major, summed left to right (GLSL mat4*vec4).
static inline f3 mul_point(const RefMat4& M, const f3& v, float w) {
    f3 r;
    r.x = M.m[0][0] * v.x + M.m[1][0] * v.y + M.m[2][0] * v.z + M.m[3][0] * w;
    r.y = M.m[0][1] * v.x + M.m[1][1] * v.y + M.m[2][1] * v.z + M.m[3][1] * w;
    r.z = M.m[0][2] * v.x + M.m[1][2] * v.y + M.m[2][2] * v.z + M.m[3][2] * w;
    return r;
}

namespace compact_detail {

struct Bounds { float lo[3], hi[3]; };
static inline Bounds emptyB() { Bounds b; for (int k = 0; k < 3; k++) { b.lo[k] = FLT_MAX; b.hi[k] = -FLT_MAX; } return b; }
static inline void grow(Bounds& b, const f3& p) {
    b.lo[0] = std::min(b.lo[0], p.x); b.lo[1] = std::min(b.lo[1], p.y); b.lo[2] = std::min(b.lo[2], p.z);
    b.hi[0] = std::max(b.hi[0], p.x); b.hi[1] = std::max(b.hi[1], p.y); b.hi[2] = std::max(b.hi[2], p.z);
}
static inline void merge(Bounds& b, const Bounds& o) {
    for (int k = 0; k < 3; k++) { b.lo[k] = std::min(b.lo[k], o.lo[k]); b.hi[k] = std::max(b.hi[k], o.hi[k]); }
}

// An internal node at the cut of a big tree: the root of a subtree that is counted and written on its own.
struct Frontier {
    uint32_t ref = 0, depth = 0;
    uint64_t triBefore = 0;              // triangles of the leaves above the cut that are visited before this subtree
    uint64_t nInt = 0, nTri = 0, nodeBase = 0, triBase = 0;
    Bounds wb = emptyB(); bool contain = true; uint32_t maxDepth = 0;
    std::string err;
};

// The depth-first walk of compact_scene over (a part of) the reference tree, in the shader's visit order.
// run<true>: validate and count (marks every node it reaches in `seen`).  run<false>: write CNodes from nodeCur and CTris
// from triCur on, with the world bounds below every node (padded leaf boxes, containment check).
struct Walk {
    const RefNode* rnodes = nullptr; uint64_t nNodes = 0, nTris = 0;
    const f3* wv = nullptr; std::atomic<uint8_t>* seen = nullptr; float padAbs = 0.f;
    CNode* nodes = nullptr; CTri* tris = nullptr;
    uint32_t cutDepth = 0xffffffffu; std::vector<Frontier>* frontier = nullptr; size_t nextFrontier = 0;
    uint64_t nodeCur = 0, triCur = 0, nInt = 0, nTri = 0;
    bool contain = true; uint32_t maxDepth = 0; Bounds rootWb = emptyB(); std::string err;

    const char* countLeaf(const RefNode& n, uint32_t& cnt) const {
        cnt = 0;
        for (int k = 0; k < 4; k++) {
            const int32_t ti = n.children[k];
            if (ti == -1) continue;
            if (ti < 0 || (uint64_t)ti >= nTris) return "leaf references triangle out of range";
            cnt++;
        }
        return nullptr;
    }
    // writes the leaf's triangles at cur.., returns the leaf code and the world bounds of the triangles
    int32_t emitLeaf(const RefNode& n, uint64_t& cur, Bounds& wb) const {
        const uint32_t first = (uint32_t)cur;
        uint32_t cnt = 0;
        wb = emptyB();
        for (int k = 0; k < 4; k++) {
            const int32_t ti = n.children[k];
            if (ti == -1) continue;
            CTri c;
            const f3 &a = wv[(uint64_t)ti * 3], &b = wv[(uint64_t)ti * 3 + 1], &cc = wv[(uint64_t)ti * 3 + 2];
            c.v0[0] = a.x; c.v0[1] = a.y; c.v0[2] = a.z;
            c.e1[0] = b.x - a.x; c.e1[1] = b.y - a.y; c.e1[2] = b.z - a.z;      // shader.comp:105
            c.e2[0] = cc.x - a.x; c.e2[1] = cc.y - a.y; c.e2[2] = cc.z - a.z;   // shader.comp:106
            c.refIndex = ti; c.pad[0] = c.pad[1] = 0;
            tris[cur++] = c;
            grow(wb, a); grow(wb, b); grow(wb, cc);
            cnt++;
        }
        // (a leaf without triangles: any non-zero `first`, so that its code differs from -1 = no child at all)
        return -(int32_t)((cnt ? first : 1u) * 8u + cnt) - 1;
    }
    void padBox(const Bounds& wb, float* lo, float* hi) const {
        for (int k = 0; k < 3; k++) {
            lo[k] = wb.lo[k] - (padAbs + 1e-5f * std::fabs(wb.lo[k]));
            hi[k] = wb.hi[k] + (padAbs + 1e-5f * std::fabs(wb.hi[k]));
        }
    }

    template <bool DRY>
    void run(uint32_t rootRef, uint32_t rootDepth) {
        // Iterative DFS.  Frame state: 0 = fresh, 1 = children.y done, 2 = both done.
        struct Frame { uint32_t ref; int32_t cidx; int state; Bounds wb; uint32_t depth; };
        std::vector<Frame> st;
        auto newInternal = [&](uint32_t ref, uint32_t depth) -> int32_t {
            Frame f; f.ref = ref; f.cidx = -1; f.state = 0; f.wb = emptyB(); f.depth = depth;
            if (DRY) nInt++;
            else {
                CNode n; std::memset(&n, 0, sizeof n);
                n.child0 = n.child1 = -1;  // empty leaf code (first 0, count 0)
                for (int k = 0; k < 3; k++) { n.lo0[k] = n.lo1[k] = FLT_MAX; n.hi0[k] = n.hi1[k] = -FLT_MAX; }
                nodes[nodeCur] = n;
                f.cidx = (int32_t)nodeCur++;
            }
            st.push_back(f);
            maxDepth = std::max(maxDepth, depth);
            return f.cidx;
        };
        newInternal(rootRef, rootDepth);
        while (!st.empty()) {
            Frame& f = st.back();
            if (f.state == 2) {
                if (!DRY) {
                    // finished: containment check of the reference box against the world bounds below it
                    const RefNode& rn = rnodes[f.ref];
                    const float rlo[3] = {rn.minLoc.x, rn.minLoc.y, rn.minLoc.z}, rhi[3] = {rn.maxLoc.x, rn.maxLoc.y, rn.maxLoc.z};
                    for (int k = 0; k < 3; k++)
                        if (f.wb.lo[k] <= f.wb.hi[k] && !(rlo[k] <= f.wb.lo[k] && f.wb.hi[k] <= rhi[k])) contain = false;
                }
                const Bounds done = f.wb;
                st.pop_back();
                if (!st.empty()) merge(st.back().wb, done); else rootWb = done;
                continue;
            }
            const int which = (f.state == 0) ? 1 : 0;   // children.y first (shader.comp:281-284 pushes x then y, pops y)
            f.state++;
            const int32_t ci = rnodes[f.ref].children[which];
            const int32_t cidx = f.cidx;
            const uint32_t depth = f.depth;
            if (ci == -1) continue;               // absent child: stays an empty leaf
            if (DRY) {
                if (ci < 0 || (uint64_t)ci >= nNodes) { err = "BVH child index out of range"; return; }
                if (seen[ci].exchange(1, std::memory_order_relaxed)) { err = "BVH node referenced twice (not a tree)"; return; }
            }
            const RefNode& cn = rnodes[ci];
            if (cn.isLeaf == 1) {
                if (DRY) {
                    uint32_t cnt;
                    if (const char* e = countLeaf(cn, cnt)) { err = e; return; }
                    nTri += cnt;
                } else {
                    Bounds wb;
                    const int32_t code = emitLeaf(cn, triCur, wb);
                    CNode& pn = nodes[cidx];
                    float* lo = which ? pn.lo1 : pn.lo0; float* hi = which ? pn.hi1 : pn.hi0;
                    if (wb.lo[0] <= wb.hi[0]) padBox(wb, lo, hi);
                    (which ? pn.child1 : pn.child0) = code;
                    merge(st.back().wb, wb);
                }
                maxDepth = std::max(maxDepth, depth + 1);
            } else if (frontier && depth + 1 >= cutDepth) {
                // the root of a subtree that is handled on its own
                if (DRY) { Frontier fr; fr.ref = (uint32_t)ci; fr.depth = depth + 1; fr.triBefore = nTri; frontier->push_back(fr); }
                else {
                    const Frontier& fr = (*frontier)[nextFrontier++];
                    CNode& pn = nodes[cidx];
                    float* lo = which ? pn.lo1 : pn.lo0; float* hi = which ? pn.hi1 : pn.hi0;
                    lo[0] = cn.minLoc.x; lo[1] = cn.minLoc.y; lo[2] = cn.minLoc.z;
                    hi[0] = cn.maxLoc.x; hi[1] = cn.maxLoc.y; hi[2] = cn.maxLoc.z;
                    (which ? pn.child1 : pn.child0) = (int32_t)fr.nodeBase;
                    merge(st.back().wb, fr.wb);
                    if (!fr.contain) contain = false;
                    maxDepth = std::max(maxDepth, fr.maxDepth);
                    triCur += fr.nTri;
                }
            } else {
                if (!DRY) {
                    CNode& pn = nodes[cidx];
                    float* lo = which ? pn.lo1 : pn.lo0; float* hi = which ? pn.hi1 : pn.hi0;
                    lo[0] = cn.minLoc.x; lo[1] = cn.minLoc.y; lo[2] = cn.minLoc.z;
                    hi[0] = cn.maxLoc.x; hi[1] = cn.maxLoc.y; hi[2] = cn.maxLoc.z;
                }
                const int32_t nidx = newInternal((uint32_t)ci, depth + 1);   // may reallocate st
                if (!DRY) (which ? nodes[cidx].child1 : nodes[cidx].child0) = nidx;
            }
        }
    }
};

}  // namespace compact_detail

// Returns "" on success, else an error message (inconsistent buffers).
static inline std::string compact_scene(const RefVertex* verts, uint64_t nVerts, const RefMat4* mats, uint64_t nMats,
                                        uint64_t nMaterials, const RefNode* rnodes, uint64_t nNodes,
                                        const RefTriangle* rtris, uint64_t nTris, CompactScene& out) {
    if (nNodes == 0) return "BVH node buffer (slot 3) is empty";
    if (nTris >= (1ull << 28)) return "too many triangles for the leaf encoding";
    // ---- per-triangle world-space vertices + shading records (independent per triangle: spread over threads) ----
    RawBuf<f3> wv;
    wv.alloc(nTris * 3);
    std::vector<float> sceneMaxAbs;
    out.shade.alloc(nTris);
    out.shadeUV.clear();
    out.nTris = nTris;
    {
        std::vector<std::string> errs(32);
        std::vector<uint64_t> errAt(32, ~0ull);
        std::vector<float> maxAbs(32, 0.f);
        const unsigned nThreads = parallel_ranges(nTris, 1u << 18, [&](uint64_t lo, uint64_t hi, unsigned k) {
            float m = 0.f;
            for (uint64_t i = lo; i < hi; i++) {
                const RefTriangle& t = rtris[i];
                const char* bad = nullptr;
                for (int v = 0; v < 3 && !bad; v++)
                    if (t.indices[v] >= nVerts) bad = ": vertex index out of range";
                if (!bad && t.meshIndex >= nMats) bad = ": meshIndex out of range";
                if (!bad && t.materialIndex >= nMaterials) bad = ": materialIndex out of range";
                if (bad) { errs[k] = "triangle " + std::to_string(i) + bad; errAt[k] = i; return; }
                const RefMat4& M = mats[t.meshIndex];
                for (int v = 0; v < 3; v++) {
                    const f3 w = mul_point(M, verts[t.indices[v]].position, 1.f);
                    wv[i * 3 + v] = w;
                    m = std::max(m, std::fabs(w.x)); m = std::max(m, std::fabs(w.y)); m = std::max(m, std::fabs(w.z));      // (a NaN never replaces m)
                }
                CShade& s = out.shade[i];
                s.pad = 0;
                const f3 P[3] = {t.p1, t.p2, t.p3};
                float* dstP[3] = {s.a, s.b, s.c};
                float* dstN[3] = {s.n1, s.n2, s.n3};
                for (int v = 0; v < 3; v++) {
                    dstP[v][0] = P[v].x; dstP[v][1] = P[v].y; dstP[v][2] = P[v].z;
                    f3 n = mul_point(M, verts[t.indices[v]].normal, 0.f);
                    dstN[v][0] = n.x; dstN[v][1] = n.y; dstN[v][2] = n.z;
                }
                s.materialIndex = t.materialIndex;
            }
            maxAbs[k] = m;
        });
        unsigned first = nThreads;                       // the error of the lowest triangle index, as the sequential loop reports it
        for (unsigned k = 0; k < nThreads; k++) if (!errs[k].empty() && (first == nThreads || errAt[k] < errAt[first])) first = k;
        if (first != nThreads) return errs[first];
        sceneMaxAbs = maxAbs;
    }

    const bool dbgT = std::getenv("RT_DEBUG") != nullptr;
    const auto tPhase0 = std::chrono::steady_clock::now();
    // ---- scene scale (for the slack terms of traversalMode 0) ----
    float scale = 0.f;
    for (float m : sceneMaxAbs) scale = std::max(scale, m);
    if (!(scale > 0.f) || !std::isfinite(scale)) scale = 1.f;
    out.sceneScale = scale;
    const float padAbs = 1e-5f * scale;

    // ---- walk the reference tree in its own visit order (children.y before children.x) ----
    // Two passes of one depth-first walk: a dry one that validates the indices, refuses anything that is not a tree and
    // counts, then one that writes into arrays of exactly that size.  A big tree is cut at kCutDepth: the subtrees below
    // the cut are counted and written by several threads (each knows where its nodes and its triangles start: triangle
    // order is the visit order, so a subtree's first index is the count of everything visited before it), the part above
    // the cut by one.
    out.nodes.clear(); out.tris.clear();
    out.maxDepth = 0;
    const RefNode& root = rnodes[0];
    out.rootLo[0] = root.minLoc.x; out.rootLo[1] = root.minLoc.y; out.rootLo[2] = root.minLoc.z;
    out.rootHi[0] = root.maxLoc.x; out.rootHi[1] = root.maxLoc.y; out.rootHi[2] = root.maxLoc.z;
    out.boxesContainTris = true;

    std::unique_ptr<std::atomic<uint8_t>[]> seen(new std::atomic<uint8_t>[nNodes]());
    compact_detail::Walk proto;
    proto.rnodes = rnodes; proto.nNodes = nNodes; proto.nTris = nTris; proto.wv = wv.data(); proto.seen = seen.get(); proto.padAbs = padAbs;

    if (root.isLeaf == 1) {
        // a lone leaf root keeps the reference's inverted-infinite box, which every ray passes (Q2)
        compact_detail::Bounds wb;
        uint32_t cnt = 0;
        if (const char* e = proto.countLeaf(root, cnt)) return e;
        out.tris.alloc(cnt);
        proto.tris = out.tris.data();
        uint64_t cur = 0;
        out.rootChild = proto.emitLeaf(root, cur, wb);
        return "";
    }

    seen[0].store(1, std::memory_order_relaxed);
    const uint32_t kCutDepth = 11;                                       // at most 1024 subtrees
    // (RT_COMPACT_PAR / RT_COMPACT_SEQ: force either form, for the test that compares them)
    const bool parallel = (nNodes >= ((uint64_t)1 << 20) || std::getenv("RT_COMPACT_PAR")) && !std::getenv("RT_COMPACT_SEQ");
    std::vector<compact_detail::Frontier> frontier;
    compact_detail::Walk top = proto;
    top.cutDepth = parallel ? kCutDepth : 0xffffffffu;
    top.frontier = &frontier;
    top.run<true>(0, 1);
    if (!top.err.empty()) return top.err;
    if (!frontier.empty()) {
        std::atomic<size_t> next{0};
        parallel_ranges(64, 1, [&](uint64_t, uint64_t, unsigned) {
            for (size_t i = next.fetch_add(1); i < frontier.size(); i = next.fetch_add(1)) {
                compact_detail::Walk w = proto;
                w.run<true>(frontier[i].ref, frontier[i].depth);
                frontier[i].nInt = w.nInt; frontier[i].nTri = w.nTri; frontier[i].err = w.err;
            }
        });
        uint64_t nodeBase = top.nInt, triSum = 0;
        for (auto& f : frontier) {
            if (!f.err.empty()) return f.err;
            f.nodeBase = nodeBase; nodeBase += f.nInt;
            f.triBase = f.triBefore + triSum; triSum += f.nTri;
        }
    }
    uint64_t totalNodes = top.nInt, totalTris = top.nTri;
    for (const auto& f : frontier) { totalNodes += f.nInt; totalTris += f.nTri; }
    if (totalNodes >= (1ull << 31)) return "too many BVH nodes";
    out.nodes.alloc(totalNodes);
    out.tris.alloc(totalTris);
    proto.nodes = out.nodes.data(); proto.tris = out.tris.data();
    if (!frontier.empty()) {
        std::atomic<size_t> next{0};
        parallel_ranges(64, 1, [&](uint64_t, uint64_t, unsigned) {
            for (size_t i = next.fetch_add(1); i < frontier.size(); i = next.fetch_add(1)) {
                compact_detail::Walk w = proto;
                w.nodeCur = frontier[i].nodeBase; w.triCur = frontier[i].triBase;
                w.run<false>(frontier[i].ref, frontier[i].depth);
                frontier[i].wb = w.rootWb; frontier[i].contain = w.contain; frontier[i].maxDepth = w.maxDepth;
            }
        });
    }
    {
        compact_detail::Walk w = proto;
        w.cutDepth = top.cutDepth; w.frontier = &frontier;
        w.run<false>(0, 1);
        out.boxesContainTris = w.contain;
        out.maxDepth = w.maxDepth;
        out.rootChild = 0;                                               // the root is the first node the top walk makes
    }

    const auto tPhase1 = std::chrono::steady_clock::now();
    // Renumber the internal nodes breadth-first: any prefix of the array is then the top of the tree, the part every ray
    // touches, which so stays one contiguous, cache-friendly run.  Triangle order (= visit rank) is untouched.
    {
        const size_t n = out.nodes.size();
        std::vector<int32_t> order; order.reserve(n);          // order[new] = old
        std::vector<int32_t> newIdx(n, -1);
        order.push_back(out.rootChild);
        for (size_t head = 0; head < order.size(); head++) {
            const CNode& nd = out.nodes[(size_t)order[head]];
            newIdx[(size_t)order[head]] = (int32_t)head;
            if (nd.child0 >= 0) order.push_back(nd.child0);
            if (nd.child1 >= 0) order.push_back(nd.child1);
        }
        if (order.size() != n) return "internal error: BVH renumbering lost nodes";
        RawBuf<CNode> re;
        re.alloc(n);
        parallel_ranges(n, 1u << 18, [&](uint64_t lo, uint64_t hi, unsigned) {
            for (size_t i = lo; i < hi; i++) {
                CNode nd = out.nodes[(size_t)order[i]];
                if (nd.child0 >= 0) nd.child0 = newIdx[(size_t)nd.child0];
                if (nd.child1 >= 0) nd.child1 = newIdx[(size_t)nd.child1];
                re[i] = nd;
            }
        });
        out.nodes.p.swap(re.p);
        out.rootChild = 0;
    }
    if (dbgT) std::fprintf(stderr, "[rt] compact_scene: tree walk %.3f s, breadth-first renumbering %.3f s\n",
                           std::chrono::duration<double>(tPhase1 - tPhase0).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - tPhase1).count());
    return "";
}

// The texture-coordinate / vertex-id records (CShadeUV) of a scene that compact_scene accepted (its index checks hold):
// only k_shade's textured and normal-mapped materials read them, so scenes without such a material never build or upload
// this table (config 4: 960 MB).
static inline void fill_shade_uv(const RefVertex* verts, const RefTriangle* rtris, uint64_t nTris, CompactScene& out) {
    out.shadeUV.alloc(nTris);
    parallel_ranges(nTris, 1u << 18, [&](uint64_t lo, uint64_t hi, unsigned) {
        for (uint64_t i = lo; i < hi; i++) {
            const RefTriangle& t = rtris[i];
            CShadeUV& su = out.shadeUV[i];
            std::memset(&su, 0, sizeof su);
            float* dstUV[3] = {su.uv1, su.uv2, su.uv3};
            for (int v = 0; v < 3; v++) { dstUV[v][0] = verts[t.indices[v]].texUV[0]; dstUV[v][1] = verts[t.indices[v]].texUV[1]; }
            su.i1 = t.indices[0]; su.i2 = t.indices[1]; su.i3 = t.indices[2];
        }
    });
}

// The two-level nodes (WNode) of a compacted tree, numbered breadth-first like the binary ones.  Returns false — and
// leaves out.wnodes empty — when the tree does not qualify: its root is a leaf, its boxes are not nested (an uploaded
// tree that is not the reference builder's: skipping a level would then skip a test the reference makes), its boxes do not
// contain their triangles, or it is so deep that three pending siblings per wide level could outgrow the 64-entry stack.
// (boxesContainTris is checked for EVERY internal reference node against the world bounds of all triangles below it
// (Walk::run, state 2), not only leaf box against triangles: a leaf two levels down whose triangles reach outside the box of
// the internal child above it clears the flag, so the clipping of such a leaf's slot to that box never culls a hit the
// reference reports — such a tree takes the reference-order walk altogether; test_two_level_nodes_only_for_trees_that_qualify.)
static inline bool derive_wide(CompactScene& out) {
    out.wnodes.clear(); out.wideDepth = 0;
    if (out.rootChild < 0 || out.nodes.empty() || !out.boxesContainTris) return false;
    const size_t nb = out.nodes.size();
    std::vector<int32_t> wideOf(nb, -1);
    std::vector<uint32_t> order, depthOf;                    // binary indices that head a wide node, breadth-first
    order.push_back((uint32_t)out.rootChild); depthOf.push_back(1);
    wideOf[(size_t)out.rootChild] = 0;
    auto inside = [](const float* lo, const float* hi, const float* plo, const float* phi) {
        for (int k = 0; k < 3; k++) if (!(plo[k] <= lo[k] && hi[k] <= phi[k])) return false;
        return true;
    };
    for (size_t head = 0; head < order.size(); head++) {
        const CNode& N = out.nodes[order[head]];
        out.wideDepth = std::max(out.wideDepth, depthOf[head]);
        const int32_t ch[2] = {N.child0, N.child1};
        for (int i = 0; i < 2; i++) {
            if (ch[i] < 0) continue;
            const CNode& C = out.nodes[(size_t)ch[i]];
            const float* plo = i ? N.lo1 : N.lo0; const float* phi = i ? N.hi1 : N.hi0;
            const int32_t g[2] = {C.child0, C.child1};
            for (int j = 0; j < 2; j++) {
                if (g[j] < 0) continue;
                const float* lo = j ? C.lo1 : C.lo0; const float* hi = j ? C.hi1 : C.hi0;
                if (!inside(lo, hi, plo, phi)) { out.wideDepth = 0; return false; }      // not nested
                wideOf[(size_t)g[j]] = (int32_t)order.size();
                order.push_back((uint32_t)g[j]); depthOf.push_back(depthOf[head] + 1);
            }
        }
    }
    if (3u * out.wideDepth + 1u > 64u) { out.wideDepth = 0; return false; }       // too deep for the 64-entry stack
    out.wnodes.resize(order.size());
    parallel_ranges(order.size(), 1u << 18, [&](uint64_t wlo, uint64_t whi, unsigned) {
    for (size_t w = wlo; w < whi; w++) {
        WNode& W = out.wnodes[w];
        std::memset(&W, 0, sizeof W);
        int ns = 0;
        auto put = [&](const float* lo, const float* hi, int32_t ref) {
            for (int k = 0; k < 3; k++) { W.box[ns][k] = lo[k]; W.box[ns][3 + k] = hi[k]; }
            W.ref[ns++] = ref;
        };
        const CNode& N = out.nodes[order[w]];
        const int32_t ch[2] = {N.child0, N.child1};
        for (int i = 0; i < 2; i++) {
            const float* plo = i ? N.lo1 : N.lo0; const float* phi = i ? N.hi1 : N.hi0;
            if (ch[i] == -1) continue;                                        // absent child
            if (ch[i] < 0) { put(plo, phi, ch[i]); continue; }                // a child that is a leaf: its padded bounds, as in the binary node
            const CNode& C = out.nodes[(size_t)ch[i]];
            const int32_t g[2] = {C.child0, C.child1};
            for (int j = 0; j < 2; j++) {
                const float* lo = j ? C.lo1 : C.lo0; const float* hi = j ? C.hi1 : C.hi0;
                if (g[j] == -1) continue;
                if (g[j] >= 0) { put(lo, hi, wideOf[(size_t)g[j]]); continue; }
                // a leaf two levels down: its padded bounds clipped to the box of the child above it, whose test this slot stands for
                float clo[3], chi[3];
                for (int k = 0; k < 3; k++) { clo[k] = std::max(lo[k], plo[k]); chi[k] = std::min(hi[k], phi[k]); }
                put(clo, chi, g[j]);
            }
        }
        for (; ns < 4; ns++) {
            for (int k = 0; k < 3; k++) { W.box[ns][k] = FLT_MAX; W.box[ns][3 + k] = -FLT_MAX; }
            W.ref[ns] = -1;
        }
    }
    });
    return true;
}

}  // namespace rt
