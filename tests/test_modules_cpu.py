"""Host-side contract of the drop-in modules (no GPU): settings dataclass, state_dict layout
against the reference's (tests/golden/state_dict_*.json), error behaviour."""
import json

import numpy as np
import pytest
import torch


def _net(C, N, K, layers, F=0):
    from randlanet.utils.modules import RandLANet, RandLANetSettings
    s = RandLANetSettings(n_classes=C, n_points=N, n_features=F, n_neighbors=K, layer_sizes=list(layers))
    return RandLANet(s, torch.device("cpu"))


@pytest.mark.parametrize("tag", ["a4", "s5", "p32", "t4"])
def test_state_dict_layout_equals_reference(golden_dir, tag):
    layout = json.load(open(f"{golden_dir}/state_dict_{tag}.json"))
    z = np.load(f"{golden_dir}/{'train_t4' if tag == 't4' else 'net_eval_' + tag}.npz")
    C, N, K, B, *layers = [int(v) for v in z["meta"]]
    sd = _net(C, N, K, layers).state_dict()
    assert [(k, list(v.shape)) for k, v in sd.items()] == [(k, list(s)) for k, s in layout]


def test_settings_contract():
    from randlanet.utils.modules import RandLANetSettings
    s = RandLANetSettings(n_classes=2)
    assert (s.n_points, s.n_features, s.n_neighbors, s.decimation) == (10000, 0, 32, 4)
    assert s.layer_sizes == [16, 64, 128, 256] and s.knn == "approximate" and s.upsampling == "nni"
    with pytest.raises(AssertionError, match="knn value"):
        RandLANetSettings(n_classes=2, knn="brute")
    with pytest.raises(AssertionError, match="upsampling value"):
        RandLANetSettings(n_classes=2, upsampling="linear")
    s.update(n_points=5, bogus=1)
    assert s.n_points == 5 and not hasattr(s, "bogus")
    from dataclasses import asdict
    assert RandLANetSettings(**json.loads(json.dumps(asdict(s)))) == s      # Model.save/load round trip


def test_same_seed_same_default_init_order():
    """Registration order follows the reference, so the parameter count per config matches."""
    net = _net(2, 40960, 16, [16, 64, 128, 256])
    assert sum(p.numel() for p in net.parameters()) == 1322666
    net = _net(13, 65536, 16, [16, 64, 128, 256, 512])
    assert sum(p.numel() for p in net.parameters()) == 5337109
    assert net._min_n_points == 4096


def test_forward_asserts_and_no_cpu_path():
    from randlanet import _hip
    net = _net(2, 2048, 16, [16, 64, 128, 256])
    with pytest.raises(AssertionError, match=r"Input should have shape \(B, N, 3 \+ F\)!"):
        net(torch.zeros(1, 2048, 4))
    with pytest.raises(AssertionError, match="at least 1024 points"):
        net(torch.zeros(1, 100, 3))
    with pytest.raises(_hip.HipKernelError, match="no CPU path"):
        net(torch.zeros(1, 2048, 3))
    with pytest.raises(_hip.HipKernelError, match="parameters only"):
        net.encoder[0].mlp1(torch.zeros(1, 8, 4, 1))
