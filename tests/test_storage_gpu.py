"""The bf16-storage throughput mode (BASELINE.json configs[1] "bf16"; rl_randlanet.h `rows_bf16`): the neighbourhood-row
tensors that live between two backward kernels are bf16 in HBM, everything that accumulates stays fp32.  The forward is
untouched by it, so loss and logits are those of the parity mode bit for bit; gradients carry bf16 rounding (2^-9 relative
per stored element) and are held against the parity mode's."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


@pytest.fixture()
def storage():
    from randlanet import _ops as ops
    before = ops.get_storage()
    yield ops
    ops.set_storage(before)


def _step(ops, mode, C, N, K, layers, B, seed=5):
    from oracle import randlanet_oracle as O
    from oracle.init_formula import formula_state_dict
    from randlanet._train import TrainStep
    from randlanet.utils.modules import RandLANet, RandLANetSettings
    ops.set_storage(mode)
    sd = formula_state_dict(O.state_dict_layout(C, 0, layers), seed=seed)
    net = RandLANet(RandLANetSettings(n_classes=C, n_points=N, n_neighbors=K, layer_sizes=list(layers)), DEV)
    net.load_state_dict(sd)
    net.fc_end[2].p = 0.0
    net.train()
    rs = np.random.RandomState(3)
    x = rs.uniform(0, 1, (B, N, 3)).astype(np.float32)
    y = np.minimum((x[..., 2] * C).astype(np.int64), C - 1)
    st = TrainStep(net, B, N, loss="dice", use_graph=False)
    st.set_batch(torch.from_numpy(x).to(DEV), torch.from_numpy(y).to(DEV))
    st.perm.copy_(torch.from_numpy(np.random.RandomState(4).permutation(N)).to(DEV))
    st._fwd_bwd()
    torch.cuda.synchronize()
    return float(st.out[0]), {n: g.detach().cpu().clone() for n, g in st.flat.grads.items()}


@pytest.mark.parametrize("C,N,K,layers,B", [
    (2, 4096, 16, [16, 64, 128, 256], 2),       # the benchmark architecture: virtual levels 0 / 1, fused d = 128, un-fused d = 256
    (5, 2048, 16, [8, 16, 32, 64], 3),          # narrow levels only (d = 8 ... 64)
])
def test_bf16_row_storage_against_the_parity_mode(storage, C, N, K, layers, B):
    ops = storage
    loss32, g32 = _step(ops, "f32", C, N, K, layers, B)
    loss16, g16 = _step(ops, "bf16", C, N, K, layers, B)
    assert ops.get_storage() == "bf16"
    assert loss16 == loss32                      # the forward does not depend on the mode
    worst, worst_name = 0.0, ""
    for name, a in g32.items():
        if (name.endswith("conv.bias") and not name.startswith("fc_end.3")) or name == "fc_start.bias":
            continue                             # in front of a BatchNorm: true gradient 0, rounding noise in both modes
        scale = float(a.abs().max())
        err = float((g16[name] - a).abs().max())
        if scale > 1e-6 and err / scale > worst:
            worst, worst_name = err / scale, name
        # a bf16-stored gradient row is off by <= 2^-9 of its value; sums over thousands of rows average that down
        assert err <= 2e-2 * scale + 2e-6, (name, err, scale)
    print(f"[bf16 storage] layers {layers}: worst gradient difference to the fp32-storage step {worst:.2e} of its scale ({worst_name})")
    assert worst > 0.0                           # the mode really took another path


def test_bf16_storage_needs_a_bf16_arithmetic_mode(storage):
    ops = storage
    mode = ops.get_wide_gemm()
    try:
        ops.set_wide_gemm("fp32")
        with pytest.raises(Exception, match="bf16 storage needs"):
            ops.set_storage("bf16")
    finally:
        ops.set_wide_gemm(mode)
    with pytest.raises(ValueError):
        ops.set_storage("fp16")
