"""CPU: the BVH oracle (oracle/bvh_ref.cpp) against the reference's recorded structure, and the product's host
builder (librt_core.so: rth_build_bvh, rth_load_scene dry run) against the oracle byte for byte."""
import numpy as np
import pytest

from oracle import gltf_flatten as G

# SURVEY.md §8 scene table: BVH nodes / leaves / depth measured with the reference's own builder
# (model.cpp:416-530 compiled verbatim in the survey container).
SURVEY_STRUCTURE = {"box": (9, 5, 5), 3: (225, 113, 10), 2: (221, 111, 11), 4: (3161, 1581, 15), 7: (28149, 14075, 21)}
SURVEY_TRIS = {"box": 10, 3: 330, 2: 342, 4: 4802, 7: 44722}


def _flat(models_root, key):
    if key == "box":
        f = G.flatten([models_root + "/box/box.gltf"])
        f["lights"], f["hasSun"] = G.lights_to_gpu(G.SCENES[3][0])
        return f
    return G.load_scene(models_root, key)


def _depth(nodes):
    best, stack = 0, [(0, 1)]
    while stack:
        i, d = stack.pop()
        best = max(best, d)
        if not nodes["isLeaf"][i]:
            for c in nodes["children"][i][:2]:
                if c >= 0:
                    stack.append((int(c), d + 1))
    return best


def masked(node_bytes):
    """Node bytes with the uninitialised pad1 (bytes 12..15 of each 48-byte record, model.h:131) removed."""
    a = np.frombuffer(node_bytes, np.uint8).reshape(-1, 48)
    return a[:, np.r_[0:12, 16:48]].tobytes()


@pytest.mark.parametrize("key", list(SURVEY_STRUCTURE))
def test_oracle_matches_reference_structure(oracle_lib, models_root, key):
    flat = _flat(models_root, key)
    assert len(flat["triangles"]) == SURVEY_TRIS[key]
    nodes = oracle_lib.build_bvh(flat["triangles"])
    n_nodes, n_leaves, depth = SURVEY_STRUCTURE[key]
    assert len(nodes) == n_nodes
    assert int(nodes["isLeaf"].sum()) == n_leaves
    assert _depth(nodes) == depth


@pytest.mark.parametrize("key", list(SURVEY_STRUCTURE))
def test_oracle_tree_invariants(oracle_lib, models_root, key):
    flat = _flat(models_root, key)
    tris = flat["triangles"]
    nodes = oracle_lib.build_bvh(tris)
    leaf = nodes["isLeaf"] == 1
    # leaves: inverted-infinite box (model.h:138-139), <= 4 triangles, every triangle exactly once
    fmax = np.float32(np.finfo(np.float32).max)
    assert (nodes["minLoc"][leaf] == fmax).all() and (nodes["maxLoc"][leaf] == -fmax).all()
    ids = nodes["children"][leaf].ravel()
    ids = ids[ids >= 0]
    assert sorted(ids.tolist()) == list(range(len(tris)))
    # internal: DFS pre-order (left child = self + 1), z/w unused, box = union of everything below
    internal = np.flatnonzero(~leaf)
    assert (nodes["children"][internal, 0] == internal + 1).all()
    assert (nodes["children"][internal, 2:] == -1).all()

    def bounds(i):
        if nodes["isLeaf"][i]:
            t = nodes["children"][i]
            t = t[t >= 0]
            return tris["minBounds"][t].min(0), tris["maxBounds"][t].max(0)
        l, r = nodes["children"][i][:2]
        (a, b), (c, d) = bounds(int(l)), bounds(int(r))
        lo, hi = np.minimum(a, c), np.maximum(b, d)
        assert np.array_equal(lo, nodes["minLoc"][i]) and np.array_equal(hi, nodes["maxLoc"][i])
        return lo, hi
    import sys
    sys.setrecursionlimit(10000)
    bounds(0)


@pytest.mark.parametrize("key", list(SURVEY_STRUCTURE))
def test_product_builder_bit_exact(oracle_lib, models_root, key):
    from raytracing_amd import scene as S
    flat = _flat(models_root, key)
    want = oracle_lib.build_bvh(flat["triangles"]).tobytes()
    got = S.build_bvh(flat["triangles"])
    assert masked(got) == masked(want)


def test_product_builder_bit_exact_synthetic_ties(oracle_lib):
    """Height-field grid: thousands of equal centroid coordinates, so the unstable std::sort tie order (Q9) decides the tree."""
    from raytracing_amd import scene as S
    n = 96
    xs, zs = np.meshgrid(np.arange(n + 1, dtype=np.float32), np.arange(n + 1, dtype=np.float32), indexing="ij")
    rng = np.random.default_rng(1234)
    ys = rng.random((n + 1, n + 1), dtype=np.float32)
    P = np.stack([xs, ys, zs], -1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
    tri_p = np.concatenate([np.stack([a, b, c], -2).reshape(-1, 3, 3), np.stack([b, d, c], -2).reshape(-1, 3, 3)])
    t = np.zeros(len(tri_p), G.TRIANGLE_DT)
    t["p1"], t["p2"], t["p3"] = tri_p[:, 0], tri_p[:, 1], tri_p[:, 2]
    t["centroidLoc"] = ((t["p1"] + t["p2"]) + t["p3"]) / np.float32(3)
    t["minBounds"] = np.minimum(np.minimum(t["p1"], t["p2"]), t["p3"])
    t["maxBounds"] = np.maximum(np.maximum(t["p1"], t["p2"]), t["p3"])
    want = oracle_lib.build_bvh(t).tobytes()
    got = S.build_bvh(t)
    assert masked(got) == masked(want)
    assert len(got) // 48 > len(t) // 4


def _tri_records(P, tri):
    t = np.zeros(len(tri), G.TRIANGLE_DT)
    t["p1"], t["p2"], t["p3"] = P[tri[:, 0]], P[tri[:, 1]], P[tri[:, 2]]
    t["centroidLoc"] = ((t["p1"] + t["p2"]) + t["p3"]) / np.float32(3)
    t["minBounds"] = np.minimum(np.minimum(t["p1"], t["p2"]), t["p3"])
    t["maxBounds"] = np.maximum(np.maximum(t["p1"], t["p2"]), t["p3"])
    return t


def test_product_builder_bit_exact_large_field_with_ties(oracle_lib):
    """320 000 triangles of config 4's generator, tied centroid keys on every axis (thousands even in y): large enough
    for what the product builder does differently from the reference's text — std::sort spread over threads (nodes above
    2^18 triangles) and chunked prefix / suffix sweeps (above 2^17).  Same bytes as the oracle, which does neither."""
    from raytracing_amd import scene as S
    from raytracing_amd import synthetic
    P, _, _, _, idx = synthetic.heightfield(400)
    t = _tri_records(P, idx.reshape(-1, 3))
    assert len(t) == 320000
    assert all(len(np.unique(t["centroidLoc"][:, a])) < len(t) for a in range(3))
    want = oracle_lib.build_bvh(t).tobytes()
    got = S.build_bvh(t)
    assert masked(got) == masked(want)


def test_product_builder_bit_exact_large_soup_without_ties(oracle_lib):
    """300 000 random triangles whose centroid keys are pairwise distinct on every axis: the product builder sorts each
    axis once for the whole scene and hands a node's order down by stable splits instead of sorting per node — the
    sorted order of distinct keys is unique, so the tree is the reference's.  One axis is then given ties (rounded keys)
    so that both kinds of axis meet in one tree."""
    from raytracing_amd import scene as S
    rng = np.random.default_rng(99)
    n = 300000
    # centroids on a quarter-unit lattice, one permutation of 0 .. n-1 per axis, vertex offsets that cancel exactly:
    # ((p1 + p2) + p3) / 3 is then the lattice point itself, and no two triangles share a coordinate
    c = np.stack([rng.permutation(n) for _ in range(3)], -1).astype(np.float32) * np.float32(0.25)
    d1 = rng.integers(-8, 9, (n, 3)).astype(np.float32) * np.float32(0.25)
    d2 = rng.integers(-8, 9, (n, 3)).astype(np.float32) * np.float32(0.25)
    p = np.stack([c + d1, c + d2, c - d1 - d2], 1).astype(np.float32)
    t = _tri_records(p.reshape(-1, 3), np.arange(3 * n).reshape(n, 3))
    assert np.array_equal(t["centroidLoc"], c)
    for rounded in (False, True):
        if rounded:
            t["centroidLoc"][:, 2] = np.round(t["centroidLoc"][:, 2] / 700)    # ~100 distinct z keys: std::sort's tie order per node
        distinct = [len(np.unique(t["centroidLoc"][:, a])) == n for a in range(3)]
        assert distinct == ([True, True, True] if not rounded else [True, True, False])
        want = oracle_lib.build_bvh(t).tobytes()
        got = S.build_bvh(t)
        assert masked(got) == masked(want), rounded


def test_degenerate_input_is_an_error_not_a_hang(oracle_lib):
    """All-identical triangles: the reference recurses without bound; both builders must stop."""
    from raytracing_amd import scene as S
    from raytracing_amd.core import RtError
    t = np.zeros(64, G.TRIANGLE_DT)
    t["p2"] = (1, 0, 0)
    t["p3"] = (0, 1, 0)
    t["centroidLoc"] = (1 / 3, 1 / 3, 0)
    t["maxBounds"] = (1, 1, 0)
    with pytest.raises(RuntimeError):
        oracle_lib.build_bvh(t)
    with pytest.raises(RtError):
        S.build_bvh(t)


def test_small_inputs(oracle_lib):
    from raytracing_amd import scene as S
    for n in (0, 1, 4, 5):
        t = np.zeros(n, G.TRIANGLE_DT)
        rng = np.random.default_rng(n)
        p = rng.random((n, 3, 3), dtype=np.float32)
        if n:
            t["p1"], t["p2"], t["p3"] = p[:, 0], p[:, 1], p[:, 2]
            t["centroidLoc"] = ((t["p1"] + t["p2"]) + t["p3"]) / np.float32(3)
            t["minBounds"] = p.min(1)
            t["maxBounds"] = p.max(1)
        want = oracle_lib.build_bvh(t).tobytes()
        got = S.build_bvh(t)
        assert masked(got) == masked(want)
        assert len(got) // 48 == (1 if n <= 4 else 3)
