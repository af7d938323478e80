"""The input-pipeline oracle (oracle/pipeline_oracle.py) against the reference's own PointCloudPreprocessor /
get_data_loader outputs (tests/golden/pipeline.npz, written by tests/golden/make_golden.py g7)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import pipeline_oracle as PO

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "pipeline.npz")), json.load(open(os.path.join(GOLDEN, "pipeline_cases.json")))


def test_preprocess_cases_match_the_reference_bit_for_bit(gold):
    z, cases = gold
    for c in cases:
        t = c["tag"]
        np.random.seed(c["seed"])
        item = PO.preprocess(z[f"{t}_xyz"], z[f"{t}_features"], z[f"{t}_labels"], c["n_sample"],
                             consistent_sampling=c["consistent"], augmentation=c["augmentation"],
                             normalization=c["normalization"])
        inp, lab = PO.collate([item])
        assert inp.dtype == np.float32 and lab.dtype == np.int64
        assert np.array_equal(inp[0], z[f"{t}_out_input"]), t
        assert np.array_equal(lab[0], z[f"{t}_out_labels"]), t
        # the global stream was consumed exactly like the reference consumed it
        assert np.array_equal(np.random.get_state()[1][:4].astype(np.int64), z[f"{t}_state_after"]), t


def test_upsampling_draws_duplicates_and_consistent_sampling_restores_the_stream():
    np.random.seed(1)
    before = np.random.get_state()[1].copy()
    idx = PO.sample_points(100, 260, consistent=True)
    assert idx.shape == (260,) and sorted(idx[:100]) == list(range(100)) and idx.max() < 100
    assert np.array_equal(np.random.get_state()[1], before)
    assert np.array_equal(idx, PO.sample_points(100, 260, consistent=True))


def test_shuffled_epoch_matches_the_reference_loader(gold):
    """Batch composition (torch's RandomSampler) and per-item stream order of one augmented epoch."""
    z, _ = gold
    ds = [(z[f"loader_xyz{i}"], np.zeros((z[f"loader_xyz{i}"].shape[0], 0), np.float32), z[f"loader_labels{i}"].astype(np.int64))
          for i in range(5)]
    torch.manual_seed(3)
    np.random.seed(4)
    # torch's DataLoader iterator first draws its base seed from the default generator; RandomSampler then draws its
    # own seed from it and permutes on a private generator
    torch.empty((), dtype=torch.int64).random_()
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    perm = torch.randperm(5, generator=g).tolist()
    aug = dict(jitter_variance=0.01, jitter_limit=0.05, scale_limit=0.2, shift_limit=0.1,
               rotation_angle_variances=(0.06, 0.06, 0.06), rotation_angle_limits=(0.18, 0.18, 0.18))
    for bi, start in enumerate(range(0, 5, 2)):
        ids = perm[start:start + 2]
        assert list(z["loader_order"][bi][:len(ids)]) == ids
        items = [PO.preprocess(*ds[i], 1024, consistent_sampling=False, augmentation=aug) for i in ids]
        inp, lab = PO.collate(items)
        assert np.array_equal(inp, z["loader_inputs"][bi][:len(ids)])
        assert np.array_equal(lab, z["loader_out_labels"][bi][:len(ids)].astype(np.int64))
