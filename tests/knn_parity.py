"""Tie-aware KNN parity check shared by the oracle tests and the GPU tests."""
import numpy as np


def check_knn(idx, d2, ref_idx, ref_d2, support, query, *, expect_lowest_index=False):
    """(i) d2 bit-equal in every slot; (ii) idx equal wherever that slot's d2 is unique in its
    row and differs from the largest d2 of the row (the k-th boundary may tie with the
    (k+1)-th, which is not in the row); (iii) elsewhere idx must be a support whose exact d2
    equals the slot's d2.  Returns the fraction of slots with equal idx."""
    idx = np.asarray(idx).astype(np.int64)
    ref_idx = np.asarray(ref_idx).astype(np.int64)
    d2 = np.asarray(d2, dtype=np.float32)
    ref_d2 = np.asarray(ref_d2, dtype=np.float32)
    assert idx.shape == ref_idx.shape and d2.shape == ref_d2.shape
    assert np.array_equal(d2.view(np.uint32), ref_d2.view(np.uint32)), "d2 not bit-identical"
    assert np.all(np.diff(d2, axis=-1) >= 0), "d2 not ascending"
    k = d2.shape[-1]
    same_as_prev = np.zeros_like(d2, dtype=bool)
    same_as_prev[..., 1:] = d2[..., 1:] == d2[..., :-1]
    same_as_next = np.zeros_like(d2, dtype=bool)
    same_as_next[..., :-1] = same_as_prev[..., 1:]
    tied = same_as_prev | same_as_next | (d2 == d2[..., -1:])
    free = ~tied
    assert np.array_equal(idx[free], ref_idx[free]), "idx differs on an untied slot"
    # every slot: the index must really be at that distance
    q = np.broadcast_to(query[..., :, None, :], idx.shape + (3,)).astype(np.float32)
    s = support[idx] if support.ndim == 2 else np.take_along_axis(
        support[:, None], idx[..., None].repeat(3, -1), axis=2)
    diff = q - s
    chk = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    assert np.array_equal(chk.astype(np.float32).view(np.uint32), d2.view(np.uint32)), \
        "idx does not point at a support with the reported d2"
    # no duplicates inside a row
    srt = np.sort(idx, axis=-1)
    assert np.all(srt[..., 1:] != srt[..., :-1]) or k == 1, "duplicate neighbour in a row"
    if expect_lowest_index:
        # inside a run of equal d2 the build orders by ascending index
        bad = same_as_prev & (idx <= np.roll(idx, 1, axis=-1))
        assert not bad.any(), "ties not in ascending index order"
    return float((idx == ref_idx).mean())
