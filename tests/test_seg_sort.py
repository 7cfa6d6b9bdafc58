"""The segmented stable radix sort behind the batched BuildLocalMap (csrc/seg_sort.h; test hook lio_seg_sort_pairs).  What it must give:
inside every segment the order of numpy's stable argsort on the sorted bits, values carried along, everything outside the segments
untouched.  The oracle's statement of the hook (std::stable_sort) is checked against numpy on the CPU; the product on the GPU against
both, over ragged and empty segments, tiles of both sizes, duplicate-heavy keys (a voxel holds ~5 points) and all-equal keys."""
import numpy as np
import pytest


def _reference(keys, vals, seg_off, seg_n, bits, passes):
    ko, vo = keys.copy(), (vals.copy() if vals is not None else np.zeros_like(keys))
    mask = np.uint32((1 << (bits * passes)) - 1) if bits * passes < 32 else np.uint32(0xFFFFFFFF)
    for o, n in zip(seg_off, seg_n):
        idx = np.argsort(keys[o:o + n] & mask, kind="stable")
        ko[o:o + n] = keys[o:o + n][idx]
        vo[o:o + n] = (vals[o:o + n][idx] if vals is not None else (o + idx).astype(np.uint32))
    return ko, vo


def _cases(rng, big):
    yield "ragged", rng.integers(0, 1 << 27, 9000, dtype=np.uint32), [0, 100, 100, 4200, 8999], [100, 0, 4000, 4700, 1], 9, 3
    yield "one element", np.array([7, 3, 9], np.uint32), [1], [1], 9, 3
    yield "all equal", np.full(5000, 12345, np.uint32), [0], [5000], 9, 3
    dup = rng.integers(0, 6000, 30000, dtype=np.uint32)               # ~5 elements per key
    yield "duplicates, 8-bit digits", dup, [0, 17000], [17000, 13000], 8, 2
    yield "top bit set sorts last", np.concatenate([rng.integers(0, 1 << 26, 3000, dtype=np.uint32), np.full(200, 0xFFFFFFFF, np.uint32)])[rng.permutation(3200)], [0], [3200], 9, 3
    if big:   # the 1024-thread tiles are chosen from 8.4 M elements per launch
        n, segs = 140000, 64
        yield "64 segments of 140 k", rng.integers(0, 1 << 25, n * segs, dtype=np.uint32), [k * n for k in range(segs)], [n - 13 * k for k in range(segs)], 9, 3


def _check(lib, rng, big):
    for name, keys, off, n, bits, passes in _cases(rng, big):
        for with_vals in (False, True):
            vals = rng.integers(0, 1 << 32, keys.shape[0], dtype=np.uint32) if with_vals else None
            ko, vo = lib.seg_sort_pairs(keys, vals, off, n, bits, passes)
            rk, rv = _reference(keys, vals, off, n, bits, passes)
            np.testing.assert_array_equal(ko, rk, err_msg=f"{name}: keys")
            inside = np.zeros(keys.shape[0], bool)
            for o, m in zip(off, n):
                inside[o:o + m] = True
            np.testing.assert_array_equal(vo[inside], rv[inside], err_msg=f"{name}: values")


def test_oracle_statement_of_the_hook(oracle):
    _check(oracle, np.random.default_rng(5), big=False)
    from lio_amd import capi

    with pytest.raises(capi.LioError):
        oracle.seg_sort_pairs(np.zeros(4, np.uint32), None, [2], [5], 9, 3)      # a segment beyond the arrays
    with pytest.raises(capi.LioError):
        oracle.seg_sort_pairs(np.zeros(4, np.uint32), None, [0], [4], 10, 3)     # digit wider than nine bits


@pytest.mark.gpu
def test_segmented_sort_on_the_gpu(hip):
    _check(hip, np.random.default_rng(6), big=True)
    from lio_amd import capi

    with pytest.raises(capi.LioError):
        hip.seg_sort_pairs(np.zeros(4, np.uint32), None, [2], [5], 9, 3)
