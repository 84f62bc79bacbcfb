"""CPU: the C++ BVH oracle (oracle/bvh_ref.cpp) and the product's builder against an independent numpy float32
re-derivation of model.cpp:417-530 (oracle/bvh_numpy.py) — byte for byte: array sweeps in numpy for the SAH cost, and a
pure-Python replay of libstdc++'s published introsort wherever tied keys make std::sort's order part of the tree (Q9).

Status of the reference-derived pins: the reference holds no fixtures, and the FNV-1a-64 node hashes quoted in
SURVEY §8c (from a probe that compiled model.cpp:416-530 in the survey container and is gone) are NOT reproduced by any
of the three builders here — box `341774c9cc8d4153` there, `2c6d751a…` / `270161c5…` here with pad bytes zeroed / skipped —
while node, leaf and depth counts all agree.  Three independently written builders agree with each other byte for byte,
so the difference lies in that probe's unrecorded input order or hashing recipe, not in a shared misreading; it stays
unexplained and the hashes are not used.  Parity of the BVH is therefore pinned by cross-derivation, not by the reference."""
import numpy as np
import pytest

from oracle import bvh_numpy as BN
from oracle import gltf_flatten as G
from tests.test_oracle_bvh import SURVEY_STRUCTURE, _depth, _flat, masked


def _soup(n, seed):
    """Random triangles with pairwise distinct centroid coordinates on every axis."""
    rng = np.random.default_rng(seed)
    t = np.zeros(n, G.TRIANGLE_DT)
    c = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
    p = c[:, None, :] + rng.uniform(-0.6, 0.6, (n, 3, 3)).astype(np.float32)
    t["p1"], t["p2"], t["p3"] = p[:, 0], p[:, 1], p[:, 2]
    t["centroidLoc"] = ((t["p1"] + t["p2"]) + t["p3"]) / np.float32(3)        # model.h:54
    t["minBounds"] = np.minimum(np.minimum(t["p1"], t["p2"]), t["p3"])
    t["maxBounds"] = np.maximum(np.maximum(t["p1"], t["p2"]), t["p3"])
    for a in range(3):
        assert len(np.unique(t["centroidLoc"][:, a])) == n
    return t


def _fnv1a64(b):
    h = 0xcbf29ce484222325
    for x in b:
        h = ((h ^ x) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.parametrize("case", ["box", "soup200", "soup3000"])
def test_three_builders_agree_byte_for_byte(oracle_lib, models_root, case):
    from raytracing_amd import scene as S
    tris = _flat(models_root, "box")["triangles"] if case == "box" else _soup(int(case[4:]), 7)
    nodes, info = BN.build(tris)
    assert not info["ties_beyond_insertion_sort"]
    want = masked(nodes.tobytes())
    assert masked(oracle_lib.build_bvh(tris).tobytes()) == want
    assert masked(S.build_bvh(tris)) == want


def test_box_tree_by_hand(models_root):
    """The 9-node tree of box.gltf (10 triangles: a cube without its top), checked against what model.cpp:417-530
    prescribes step by step rather than against another builder."""
    tris = _flat(models_root, "box")["triangles"]
    nodes, info = BN.build(tris)
    assert len(nodes) == 9 and info["leaves"] == 5 and info["depth"] == 5
    leaf = nodes["isLeaf"] == 1
    # pre-order: every internal node's left child is the next record; its right child follows the whole left subtree
    def size(i):
        return 1 if leaf[i] else 1 + size(nodes["children"][i][0]) + size(nodes["children"][i][1])
    for i in np.flatnonzero(~leaf):
        l, r = nodes["children"][i][:2]
        assert l == i + 1 and r == l + size(l)
    # the root box is the mesh's bounds; leaves carry the inverted sentinels and together hold each triangle once
    pts = np.concatenate([tris["p1"], tris["p2"], tris["p3"]])
    assert np.array_equal(nodes["minLoc"][0], pts.min(0)) and np.array_equal(nodes["maxLoc"][0], pts.max(0))
    assert (nodes["minLoc"][leaf] == BN.FMAX).all() and (nodes["maxLoc"][leaf] == -BN.FMAX).all()
    held = nodes["children"][leaf].ravel()
    assert sorted(held[held >= 0].tolist()) == list(range(10))
    # the root split, recomputed here term by term for the winning axis: cost(i) = 1 + SA(prefix_i)/SA(all) * i + SA(suffix_i)/SA(all) * (n - i)
    # with prefix_i INCLUDING triangle i (model.cpp:433-453), first strict minimum, axes compared in order x, y, z
    n = len(tris)
    best = None
    for axis in range(3):
        order = np.argsort(tris["centroidLoc"][:, axis], kind="stable")      # 10 <= 16 elements: std::sort is an insertion sort, stable
        lo, hi = tris["minBounds"][order], tris["maxBounds"][order]
        for i in range(n):
            plo, phi = lo[:i + 1].min(0), hi[:i + 1].max(0)
            slo, shi = lo[i:].min(0), hi[i:].max(0)
            alo, ahi = np.minimum(plo, slo), np.maximum(phi, shi)
            sa = lambda a, b: np.float32(2) * (((b - a)[0] * (b - a)[1] + (b - a)[1] * (b - a)[2]) + (b - a)[2] * (b - a)[0])
            cost = np.float32(1) + ((sa(plo, phi) / sa(alo, ahi)) * np.float32(i) + (sa(slo, shi) / sa(alo, ahi)) * np.float32(n - i))
            if best is None or cost < best[0]:
                best = (cost, axis, i, order)
    _, axis, split, order = best
    left_ids = set(order[:split].tolist())
    def tris_below(i):
        if leaf[i]:
            c = nodes["children"][i]
            return set(c[c >= 0].tolist())
        return tris_below(nodes["children"][i][0]) | tris_below(nodes["children"][i][1])
    assert tris_below(nodes["children"][0][0]) == left_ids and 0 < split < n
    # leaf slots keep the winning sorted order of their parent (convertToGPU reads box->triangles[k] in order)
    first_leaf = int(np.flatnonzero(leaf)[0])
    c = nodes["children"][first_leaf]
    assert all(int(x) in left_ids for x in c[c >= 0])


def test_std_sort_replay_is_a_sort_and_differs_from_a_stable_one():
    """The Python replay of libstdc++'s introsort sorts, and on tied keys it leaves an order a stable sort would not —
    which is why the tie order is part of the tree (Q9)."""
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 12, 400).astype(np.float32)
    order = BN.std_sort_order(keys)
    assert sorted(order.tolist()) == list(range(400))
    assert (np.diff(keys[order]) >= 0).all()
    assert not np.array_equal(order, np.argsort(keys, kind="stable"))
    small = rng.integers(0, 3, 16).astype(np.float32)          # <= 16 elements: insertion sort only, stable
    assert np.array_equal(BN.std_sort_order(small), np.argsort(small, kind="stable"))


@pytest.mark.parametrize("key", [2, 3, 4, 7])
def test_bundled_scenes_against_the_numpy_builder(oracle_lib, models_root, key):
    """Scenes whose nodes of more than 16 triangles have tied centroid keys (walls, floors, axis-aligned quads): the numpy
    builder replays libstdc++'s introsort there.  All three builders: the same bytes, and the structure the survey
    recorded from the reference's own builder."""
    from raytracing_amd import scene as S
    tris = _flat(models_root, key)["triangles"]
    nodes, info = BN.build(tris)
    assert info["ties_beyond_insertion_sort"]                   # these scenes do exercise the tie order
    want = oracle_lib.build_bvh(tris)
    assert masked(nodes.tobytes()) == masked(want.tobytes())
    assert masked(S.build_bvh(tris)) == masked(want.tobytes())
    assert (len(nodes), info["leaves"], info["depth"]) == SURVEY_STRUCTURE[key]
    assert _depth(want) == SURVEY_STRUCTURE[key][2]


def test_survey_hashes_are_not_reproduced(oracle_lib, models_root):
    """Documents the open item instead of hiding it: the SURVEY §8c hash of the box tree differs from the hash of the
    bytes all three builders agree on, under both treatments of the uninitialised pad word."""
    tris = _flat(models_root, "box")["triangles"]
    raw = oracle_lib.build_bvh(tris).tobytes()
    zeroed = np.frombuffer(raw, np.uint8).reshape(-1, 48).copy()
    zeroed[:, 12:16] = 0
    ours = {_fnv1a64(masked(raw)), _fnv1a64(zeroed.tobytes())}
    assert 0x341774c9cc8d4153 not in ours
