"""ORACLE — test infrastructure only (tests/ may import it; the product path never does).

A SECOND, independent restatement of the reference's SAH builder, written from the reference text alone
(/root/reference/Raytracing/model.cpp:417-530, model.h:81-118,137-141) in numpy float32 — array sweeps instead of the
pointer loops of oracle/bvh_ref.cpp and of the product's csrc/host/bvh_build.h — so that a misreading shared by those
two (flatten order, which side owns element i, leaf slot order, first-minimum rule) shows up as a byte difference.

Sorting: the reference sorts with std::sort, whose order among EQUAL keys is whatever the library's introsort does to
that input (Q9: part of the tree).  Any sort gives the same order when all keys differ, and libstdc++ runs a plain
insertion sort — stable — for ranges of at most 16 elements: there a stable argsort is used.  For a range of more than
16 elements with tied keys, `std_sort_order` below replays libstdc++'s published algorithm (bits/stl_algo.h: introsort
loop with median-of-three to the front and unguarded Hoare partition down to 16 elements, then one insertion sort) on the
index list in pure Python, including its heapsort fallback (bits/stl_heap.h) for recursion deeper than 2 log2 n.
`info["ties_beyond_insertion_sort"]` says whether that replay was needed at all.
"""
import numpy as np

FMAX = np.float32(np.finfo(np.float32).max)
NODE_DT = np.dtype({"names": ["minLoc", "pad1", "maxLoc", "isLeaf", "children"],
                    "formats": [(np.float32, 3), np.float32, (np.float32, 3), np.uint32, (np.int32, 4)],
                    "offsets": [0, 12, 16, 28, 32], "itemsize": 48})


def _surface_area(lo, hi):
    """BoundingBox::calcSurfaceArea (model.h:115-118): 2 (ex ey + ey ez + ez ex), float32, left to right."""
    e = (hi - lo).astype(np.float32)
    return np.float32(2.0) * ((e[..., 0] * e[..., 1] + e[..., 1] * e[..., 2]) + e[..., 2] * e[..., 0])


def std_sort_order(keys):
    """The permutation libstdc++'s std::sort(first, last, key-only '<') leaves, for a list of float keys."""
    k = [float(x) for x in keys]
    a = list(range(len(k)))                       # a[pos] = original index; comparisons read k[a[pos]]
    lt = lambda x, y: k[x] < k[y]

    def unguarded_linear_insert(last):
        val = a[last]
        nxt = last - 1
        while lt(val, a[nxt]):
            a[last] = a[nxt]
            last = nxt
            nxt -= 1
        a[last] = val

    def insertion_sort(first, last):
        for i in range(first + 1, last):
            if lt(a[i], a[first]):
                val = a[i]
                a[first + 1:i + 1] = a[first:i]
                a[first] = val
            else:
                unguarded_linear_insert(i)

    def move_median_to_first(result, x, y, z):
        if lt(a[x], a[y]):
            if lt(a[y], a[z]): w = y
            elif lt(a[x], a[z]): w = z
            else: w = x
        elif lt(a[x], a[z]): w = x
        elif lt(a[y], a[z]): w = z
        else: w = y
        a[result], a[w] = a[w], a[result]

    def unguarded_partition(first, last, pivot):
        while True:
            while lt(a[first], a[pivot]):
                first += 1
            last -= 1
            while lt(a[pivot], a[last]):
                last -= 1
            if not first < last:
                return first
            a[first], a[last] = a[last], a[first]
            first += 1

    # bits/stl_heap.h: __push_heap, __adjust_heap, __make_heap, __pop_heap, __sort_heap on a[first:first+len]
    def push_heap(first, hole, top, value):
        parent = (hole - 1) // 2 if hole > 0 else 0
        while hole > top and lt(a[first + parent], value):
            a[first + hole] = a[first + parent]
            hole = parent
            parent = (hole - 1) // 2 if hole > 0 else 0
        a[first + hole] = value

    def adjust_heap(first, hole, length, value):
        top = hole
        child = hole
        while child < (length - 1) // 2:
            child = 2 * (child + 1)
            if lt(a[first + child], a[first + child - 1]):
                child -= 1
            a[first + hole] = a[first + child]
            hole = child
        if (length & 1) == 0 and child == (length - 2) // 2:
            child = 2 * (child + 1)
            a[first + hole] = a[first + child - 1]
            hole = child - 1
        push_heap(first, hole, top, value)

    def heap_sort(first, last):                   # std::partial_sort(first, last, last): __heap_select (= make_heap) + __sort_heap
        length = last - first
        if length >= 2:
            parent = (length - 2) // 2
            while True:
                adjust_heap(first, parent, length, a[first + parent])
                if parent == 0:
                    break
                parent -= 1
        while last - first > 1:
            last -= 1
            value = a[last]
            a[last] = a[first]
            adjust_heap(first, 0, last - first, value)

    def introsort_loop(first, last, depth):
        while last - first > 16:
            if depth == 0:
                heap_sort(first, last)
                return
            depth -= 1
            mid = first + (last - first) // 2
            move_median_to_first(first, first + 1, mid, last - 1)
            cut = unguarded_partition(first + 1, last, first)
            introsort_loop(cut, last, depth)
            last = cut

    n = len(a)
    if n:
        introsort_loop(0, n, 2 * (n.bit_length() - 1))
        if n > 16:
            insertion_sort(0, 16)
            for i in range(16, n):
                unguarded_linear_insert(i)
        else:
            insertion_sort(0, n)
    return np.array(a, np.int64)


def _axis_split(tmin, tmax, keys):
    """computeForAxis (model.cpp:417-462) for one axis.  Returns (bestCost, bestIndex, order, tied)."""
    order = np.argsort(keys, kind="stable")
    k = keys[order]
    tied = bool((k[1:] == k[:-1]).any())
    if tied and len(keys) > 16:
        order = std_sort_order(keys)
    lo, hi = tmin[order], tmax[order]
    n = len(order)
    pre_lo, pre_hi = np.minimum.accumulate(lo, 0), np.maximum.accumulate(hi, 0)            # prefix[i] includes element i
    suf_lo, suf_hi = np.minimum.accumulate(lo[::-1], 0)[::-1], np.maximum.accumulate(hi[::-1], 0)[::-1]
    parent = _surface_area(np.minimum(pre_lo, suf_lo), np.maximum(pre_hi, suf_hi))
    i = np.arange(n)
    with np.errstate(all="ignore"):
        left = (_surface_area(pre_lo, pre_hi) / parent) * i.astype(np.float32)
        right = (_surface_area(suf_lo, suf_hi) / parent) * (n - i).astype(np.float32)     # size_t -> float
        cost = np.float32(1.0) + (left + right)
    best_cost, best_index = FMAX, n // 2                                                   # model.cpp:445-446
    below = np.flatnonzero(cost < FMAX)
    if len(below):
        m = cost[below].min()
        best_index = int(below[np.flatnonzero(cost[below] == m)[0]])                       # first strict minimum
        best_cost = cost[best_index]
    return best_cost, best_index, order, tied


def build(triangles, leaf_size=4):
    """buildBVH + convertToGPU.  triangles: structured array with p1, p2, p3, centroidLoc, minBounds, maxBounds.
    Returns (nodes [NODE_DT, DFS pre-order], info dict)."""
    p = np.stack([triangles["p1"], triangles["p2"], triangles["p3"]], 1).astype(np.float32)
    cen = np.ascontiguousarray(triangles["centroidLoc"], np.float32)
    tmin = np.ascontiguousarray(triangles["minBounds"], np.float32)
    tmax = np.ascontiguousarray(triangles["maxBounds"], np.float32)
    out = []
    info = {"ties_beyond_insertion_sort": False, "leaves": 0, "depth": 0}

    def rec(ids, depth):
        """Returns (index in out, host box lo, host box hi).  ids in the parent's winning sorted order."""
        me = len(out)
        out.append(None)
        info["depth"] = max(info["depth"], depth)
        n = len(ids)
        if n <= leaf_size:                                       # model.cpp:472-473, model.h:94-108
            pts = p[ids].reshape(-1, 3)
            lo = pts.min(0) if n else np.full(3, FMAX, np.float32)
            hi = pts.max(0) if n else np.full(3, -FMAX, np.float32)
            rec_ = np.zeros((), NODE_DT)
            rec_["minLoc"], rec_["maxLoc"], rec_["isLeaf"] = FMAX, -FMAX, 1     # a leaf has no children: both sentinels (model.h:138-139)
            ch = np.full(4, -1, np.int32)
            ch[:n] = ids
            rec_["children"] = ch
            out[me] = rec_
            info["leaves"] += 1
            return me, lo, hi
        best = None
        for axis in range(3):                                    # strict < in axis order 0, 1, 2 (model.cpp:488-492)
            cost, split, order, tied = _axis_split(tmin[ids], tmax[ids], cen[ids, axis])
            if tied and n > 16:
                info["ties_beyond_insertion_sort"] = True
            if best is None or cost < best[0]:
                best = (cost, split, order)
        _, split, order = best
        if split == 0 or split == n:
            raise RuntimeError("degenerate split: the reference recurses without bound here (model.cpp:502-508)")
        srt = ids[order]
        li, llo, lhi = rec(srt[:split], depth + 1)
        ri, rlo, rhi = rec(srt[split:], depth + 1)
        lo, hi = np.minimum(llo, rlo), np.maximum(lhi, rhi)      # model.h:85-90 / 138-139
        rec_ = np.zeros((), NODE_DT)
        rec_["minLoc"], rec_["maxLoc"], rec_["isLeaf"] = lo, hi, 0
        rec_["children"] = (li, ri, -1, -1)
        out[me] = rec_
        return me, lo, hi

    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))
    rec(np.arange(len(triangles)), 1)
    return np.array(out, NODE_DT), info
