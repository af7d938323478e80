"""Pins oracle/knn_oracle.c against the reference's own C++ KNN through the committed golden
vectors (tests/golden/knn_cases.npz, produced by knn_tpk.knn - see make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import randlanet_oracle as O
from knn_parity import check_knn


def _cases(golden_dir):
    z = np.load(f"{golden_dir}/knn_cases.npz")
    tags = sorted({k.split("/")[0] for k in z.files if k.endswith("/idx")})
    for t in tags:
        data = str(z[f"{t}/data"])
        sup = z[f"{data}/support"]
        qry = z[f"{data}/query"] if f"{data}/query" in z.files else sup
        yield t, sup, qry, int(z[f"{t}/k"]), z[f"{t}/idx"], z[f"{t}/d2"]


@pytest.mark.parametrize("method", ["brute", "grid"])
def test_oracle_matches_reference_golden(golden_dir, method):
    n = 0
    for tag, sup, qry, k, ref_idx, ref_d2 in _cases(golden_dir):
        idx, d2 = O.knn(torch.from_numpy(sup[None]), torch.from_numpy(qry[None]), k, method)
        frac = check_knn(idx[0].numpy(), d2[0].numpy(), ref_idx, ref_d2, sup, qry,
                         expect_lowest_index=True)
        if tag.startswith("uniform") or tag.startswith("cross"):
            assert frac == 1.0, tag
        n += 1
    assert n >= 14


def test_grid_equals_brute_exactly():
    rs = np.random.RandomState(5)
    for (ns, nq, k) in [(17, 40, 17), (500, 300, 8), (3000, 3000, 16), (33, 1, 1)]:
        s = torch.from_numpy(rs.normal(0, 1, (2, ns, 3)).astype(np.float32))
        q = torch.from_numpy(rs.normal(0, 2, (2, nq, 3)).astype(np.float32))
        i1, d1 = O.knn(s, q, k, "brute")
        i2, d2 = O.knn(s, q, k, "grid")
        assert torch.equal(i1, i2) and torch.equal(d1, d2)
    # degenerate: all points identical / collinear
    s = torch.zeros(1, 50, 3)
    i1, d1 = O.knn(s, s, 5, "grid")
    assert torch.equal(i1[0, 7], torch.arange(5)) and float(d1.abs().max()) == 0.0
    line = torch.zeros(1, 200, 3)
    line[0, :, 0] = torch.arange(200).float()
    i1, d1 = O.knn(line, line, 3, "brute")
    i2, d2 = O.knn(line, line, 3, "grid")
    assert torch.equal(i1, i2) and torch.equal(d1, d2)


def test_error_contract():
    s = torch.zeros(1, 3, 3)
    with pytest.raises(RuntimeError, match="Not enough points"):
        O.knn(s, s, 4)                       # knn.cpp:15-17
    with pytest.raises(RuntimeError, match="contiguous"):
        O.knn(torch.zeros(1, 3, 8)[..., ::2][..., :3], s, 1)
