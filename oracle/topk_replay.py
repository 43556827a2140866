"""Test infrastructure (oracle): torch.topk's CPU selection restated step by step, ties included.

The reference builds the GNN's neighbour graph with `dist.topk(k+1, largest=False, sorted=True)` on a dense fp32
distance matrix (ref:models/gcn.py:48-49).  Which of several EXACTLY equal distances survive the cut -- and in which
order equal entries come out -- is decided by the implementation: PyTorch's CPU kernel (third-party, not vendored in the
reference; pinned here: torch 2.10, aten/src/ATen/native/cpu/TopKImpl.h `topk_impl_loop`) fills a vector of
(value, index) pairs and runs, with the comparator `x.first < y.first` (NaN last),

    k * 64 <= n :  std::partial_sort(begin, begin + k, end)
    else        :  std::nth_element(begin, begin + k - 1, end); std::sort(begin, begin + k - 1)

on libstdc++ (GCC 11).  Both are restated below from the published libstdc++ algorithms (bits/stl_algo.h,
bits/stl_heap.h): __heap_select / __make_heap / __adjust_heap / __push_heap / __sort_heap, and __introselect /
__move_median_to_first / __unguarded_partition / __insertion_sort.  Pinned by tests/test_oracle_model.py against
torch.topk itself on rows full of equal values, in both regimes.  Pure-Python loops: small cases only."""


def _adjust_heap(a, first, hole, length, value, less):
    top = hole
    child = hole
    while child < (length - 1) // 2:
        child = 2 * (child + 1)
        if less(a[first + child], a[first + child - 1]):
            child -= 1
        a[first + hole] = a[first + child]
        hole = child
    if (length & 1) == 0 and child == (length - 2) // 2:
        child = 2 * (child + 1)
        a[first + hole] = a[first + child - 1]
        hole = child - 1
    parent = (hole - 1) // 2                      # __push_heap
    while hole > top and less(a[first + parent], value):
        a[first + hole] = a[first + parent]
        hole = parent
        parent = (hole - 1) // 2
    a[first + hole] = value


def _make_heap(a, first, last, less):
    length = last - first
    if length < 2:
        return
    parent = (length - 2) // 2
    while True:
        _adjust_heap(a, first, parent, length, a[first + parent], less)
        if parent == 0:
            return
        parent -= 1


def _pop_heap(a, first, last, result, less):
    value = a[result]
    a[result] = a[first]
    _adjust_heap(a, first, 0, last - first, value, less)


def _heap_select(a, first, middle, last, less):
    _make_heap(a, first, middle, less)
    for i in range(middle, last):
        if less(a[i], a[first]):
            _pop_heap(a, first, middle, i, less)


def _sort_heap(a, first, last, less):
    while last - first > 1:
        last -= 1
        _pop_heap(a, first, last, last, less)


def _insertion_sort(a, first, last, less):
    if first == last:
        return
    for i in range(first + 1, last):
        v = a[i]
        if less(v, a[first]):
            a[first + 1:i + 1] = a[first:i]
            a[first] = v
        else:                                      # __unguarded_linear_insert
            j = i
            while less(v, a[j - 1]):
                a[j] = a[j - 1]
                j -= 1
            a[j] = v


def _median_to_first(a, result, x, y, z, less):
    if less(a[x], a[y]):
        if less(a[y], a[z]):
            pick = y
        elif less(a[x], a[z]):
            pick = z
        else:
            pick = x
    elif less(a[x], a[z]):
        pick = x
    elif less(a[y], a[z]):
        pick = z
    else:
        pick = y
    a[result], a[pick] = a[pick], a[result]


def _unguarded_partition_pivot(a, first, last, less):
    mid = first + (last - first) // 2
    _median_to_first(a, first, first + 1, mid, last - 1, less)
    lo, hi, pivot = first + 1, last, first
    while True:
        while less(a[lo], a[pivot]):
            lo += 1
        hi -= 1
        while less(a[pivot], a[hi]):
            hi -= 1
        if not lo < hi:
            return lo
        a[lo], a[hi] = a[hi], a[lo]
        lo += 1


def _introselect(a, first, nth, last, depth, less):
    while last - first > 3:
        if depth == 0:
            _heap_select(a, first, nth + 1, last, less)
            a[first], a[nth] = a[nth], a[first]
            return
        depth -= 1
        cut = _unguarded_partition_pivot(a, first, last, less)
        if cut <= nth:
            first = cut
        else:
            last = cut
    _insertion_sort(a, first, last, less)


def topk_smallest_indices(values, k):
    """Indices torch.topk(values, k, largest=False, sorted=True) returns for one row on the CPU (finite values)."""
    n = len(values)
    a = [(float(v), j) for j, v in enumerate(values)]

    def less(x, y):
        return x[0] < y[0]

    if k * 64 <= n:
        _heap_select(a, 0, k, n, less)
        _sort_heap(a, 0, k, less)
    else:
        if n > 0 and k >= 1:
            depth = 2 * (n.bit_length() - 1)       # std::__lg(n) * 2
            _introselect(a, 0, k - 1, n, depth, less)
        # std::sort(begin, begin + k - 1): introsort; for k - 1 <= 16 elements it is the insertion sort alone
        if k - 1 > 16:
            raise NotImplementedError("restated for k - 1 <= 16 (the path uses k = 11)")
        _insertion_sort(a, 0, max(k - 1, 0), less)
    return [a[j][1] for j in range(k)]
