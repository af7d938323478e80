"""BASELINE.json's configurations at their full sizes on the MI355X: logits of the HIP path against the
oracle (CPU restatement, pinned to the reference by tests/golden) on the same seeded cloud and weights,
plus a train-mode step for the K=32 setting of train.py (the path without the fused 16-neighbour kernels)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _pair(C, N, K, layers, seed):
    from oracle import randlanet_oracle as O
    from oracle.init_formula import formula_state_dict
    from randlanet.utils.modules import RandLANet, RandLANetSettings
    sd = formula_state_dict(O.state_dict_layout(C, 0, layers), seed=seed)
    net = RandLANet(RandLANetSettings(n_classes=C, n_points=N, n_neighbors=K, layer_sizes=list(layers)), DEV)
    net.load_state_dict(sd)
    return net, sd


@pytest.mark.parametrize("tag,C,N,K,layers", [
    ("A", 2, 40960, 16, [16, 64, 128, 256]),             # configs[1]/[2]: the benchmark architecture
    ("S", 13, 65536, 16, [16, 64, 128, 256, 512]),       # configs[3]: S3DIS-shaped, 5 encoder layers
    ("Kt", 20, 122880, 16, [16, 64, 128, 256]),          # configs[4]: SemanticKITTI-shaped
])
def test_full_size_eval_logits_match_oracle(tag, C, N, K, layers):
    from oracle import randlanet_oracle as O
    net, sd = _pair(C, N, K, layers, seed=17)
    net.eval()
    rs = np.random.RandomState(5)
    x = rs.uniform(0, 1, (1, N, 3)).astype(np.float32)
    np.random.seed(8)
    perm = np.random.permutation(N)
    np.random.seed(8)
    with torch.no_grad():
        logits = net(torch.from_numpy(x).to(DEV)).cpu()
        ref = O.forward(sd, torch.from_numpy(x), perm, layer_sizes=layers, n_neighbors=K)
    assert logits.shape == (1, C, N)
    err = (logits - ref).abs()
    # north_star: per-point logits within 1e-3 of the CPU forward on the same cloud
    print(f"[parity] config {tag}: max |logit - oracle| = {float(err.max()):.3e} (logit range {float(ref.abs().max()):.2f})")
    assert float(err.max()) < 1e-3, (tag, float(err.max()))
    assert torch.equal(logits.argmax(1), ref.argmax(1)) or float((logits.argmax(1) != ref.argmax(1)).float().mean()) < 1e-4


def _oracle_step(sd, x, y, perm, layers, K, dtype=torch.float32, wide_noise=0.0, noise_seed=0):
    """The oracle's train-mode forward + dice + autograd in `dtype` (float64 = the yardstick the fp32 oracle's own rounding
    is measured against): (logits, loss, {name: gradient}).  wide_noise > 0: every output of a layer wider than 64 channels
    (the layers the bf16x3 kernels compute) gets independent relative noise of that size (x its rms), and so does the
    gradient arriving at it in the backward - an exact evaluation of the function with the products of those layers
    (forward, input gradient, weight gradient) perturbed the way a 2^-17-per-product arithmetic perturbs them."""
    from oracle import randlanet_oracle as O
    from oracle.loss_metrics_oracle import loss_by_name
    P = {k: (v.to(dtype).clone().requires_grad_(True) if v.is_floating_point() and "running" not in k
             else (v.to(dtype) if v.is_floating_point() else v.clone())) for k, v in sd.items()}
    conv2d, linear, convT = O.F.conv2d, O.F.linear, O.F.conv_transpose2d
    gen = torch.Generator().manual_seed(noise_seed)

    class Noise(torch.autograd.Function):
        """y = x + noise in the forward, and the same relative noise on the gradient coming back: the layer's forward
        product, and its input / weight gradient products, all computed with a relative error of `wide_noise`."""
        @staticmethod
        def forward(ctx, t):
            return t + wide_noise * t.pow(2).mean().sqrt() * torch.randn(t.shape, generator=gen, dtype=t.dtype)

        @staticmethod
        def backward(ctx, g):
            return g + wide_noise * g.pow(2).mean().sqrt() * torch.randn(g.shape, generator=gen, dtype=g.dtype)

    def noisy(fn):
        def f(inp, w, b=None, *a, **k):
            out = fn(inp, w, b, *a, **k)
            # layers the bf16x3 kernels compute: everything wider than 64 channels, and every attention score Linear
            # (d x d, d >= 16: with 16 neighbours those run inside the fused tile kernels, bf16x3 as well)
            if max(w.shape[0], w.shape[1]) > 64 or (fn is linear and w.shape[1] >= 16):
                out = Noise.apply(out)
            return out
        return f
    if wide_noise > 0.0:
        O.F.conv2d, O.F.linear, O.F.conv_transpose2d = noisy(conv2d), noisy(linear), noisy(convT)
    try:
        ref = O.forward(P, torch.from_numpy(x).to(dtype), perm, layer_sizes=layers, n_neighbors=K, training=True, dropout_p=0.0)
        loss = loss_by_name("dice", ref, torch.from_numpy(y))
        loss.backward()
    finally:
        O.F.conv2d, O.F.linear, O.F.conv_transpose2d = conv2d, linear, convT
    return ref.detach(), float(loss.detach()), {k: v.grad for k, v in P.items() if v.requires_grad}, P


def _yardstick(tag, mode, grads_hip, g32, g64):
    """The gradient bound stated against an fp64 evaluation of the same function: per tensor,
        |g_hip - g_64| <= YARD[mode] * |g_yard - g_64| + 1e-5 * scale + 1e-8     (max norms)
    where g_yard is, for the exact-product mode, the fp32 CPU oracle (same unit round-off: its own distance from fp64 says
    how ill-conditioned the test point is), and for bf16x3 (2^-17 per product; the kernels measure 2-5e-5 max / ~5e-6 rms
    of the output's rms, tools/precision_probe.py) an EXACT fp64 evaluation whose wide-layer outputs carry 5e-6 relative
    noise, forward and backward (_oracle_step(wide_noise=5e-6), worst of two draws).  The response of these random-weight points
    to such noise is far from linear (K = 32 point, fp64: noise 1e-7 -> 6e-7 of a gradient's scale, 1e-6 -> 2e-3,
    1e-5 -> 0.5), which is why a multiple of the fp32 oracle's distance cannot serve for both modes."""
    worst, worst_name, cond, table = 0.0, "", 0.0, []
    for name, g in grads_hip.items():
        r64 = g64[name]
        scale = float(r64.abs().max())
        e_hip = float((g.double() - r64).abs().max())
        e_32 = float((g32[name].double() - r64).abs().max())
        floor = 1e-5 * scale + 1e-8            # (a bias in front of a BatchNorm: true gradient 0, see _zero_gradient)
        if _zero_gradient(name):
            continue
        if scale > 1e-12:
            cond = max(cond, e_32 / scale)
        ratio = e_hip / max(e_32, 1e-30)
        table.append((ratio if e_hip > floor else 0.0, name, e_hip / max(scale, 1e-30), e_32 / max(scale, 1e-30)))
        if e_hip > floor and ratio > worst:
            worst, worst_name = ratio, name
    for ratio, name, rh, r32 in sorted(table, reverse=True)[:6]:
        print(f"    {name:42s} |g_hip-g64|/scale {rh:.2e}   |g_oracle32-g64|/scale {r32:.2e}   ratio {ratio:9.1f}")
    for name, g in grads_hip.items():
        if _zero_gradient(name):
            continue
        r64 = g64[name]
        scale = float(r64.abs().max())
        e_hip = float((g.double() - r64).abs().max())
        e_32 = float((g32[name].double() - r64).abs().max())
        assert e_hip <= YARD[mode] * e_32 + 1e-5 * scale + 1e-8, (tag, mode, name, e_hip, e_32, scale)
    print(f"[fp64 yardstick] {tag} {mode}: worst |g_hip - g64| / |g_oracle32 - g64| = {worst:.2f} ({worst_name or 'all at the floor'}); "
          f"bound {YARD[mode]:g}; the yardstick itself sits up to {cond:.1e} (relative) from fp64")
    return worst


def _noisy_yard(sd, x, y, perm, layers, K, g64):
    """Per tensor, the draw (of two) of the noisy fp64 evaluation that lies further from the clean one."""
    a = _oracle_step(sd, x, y, perm, layers, K, torch.float64, WIDE_NOISE, 1)[2]
    b = _oracle_step(sd, x, y, perm, layers, K, torch.float64, WIDE_NOISE, 2)[2]
    return {k: (a[k] if float((a[k] - g64[k]).abs().max()) >= float((b[k] - g64[k]).abs().max()) else b[k]) for k in g64}


def _zero_gradient(name):
    """A conv bias in front of a BatchNorm cancels in (y - mean): its true gradient is exactly 0 and what either side holds
    is rounding noise of sums over up to 5 M rows.  fc_start (a Linear in front of bn_start, modules.py:495-500) is the sixth
    such parameter: in fp64 its gradient is 1e-16, ours sat at 0.98e-8 / 1.00e-8 either side of the 1e-8
    floor below depending on how the statistics' partial sums are grouped."""
    return (name.endswith("conv.bias") and not name.startswith("fc_end.3")) or name == "fc_start.bias"


def _check_gradient(ctx, name, g, r, bound):
    e, scale = float((g - r).abs().max()), float(r.abs().max())
    if _zero_gradient(name):
        # both are noise around 0: ours must not be noisier than a few times the oracle's
        assert float(g.abs().max()) <= 4.0 * scale + 2e-5, ctx + (name, float(g.abs().max()), scale)
        return 0.0
    assert e < bound * scale + 2e-5, ctx + (name, e, scale)
    return e / scale if scale > 1e-4 else 0.0


@pytest.mark.parametrize("tag,C,N,K,layers,B", [
    ("A", 2, 40960, 16, [16, 64, 128, 256], 4),          # BASELINE config A: 4 clouds per GPU
    ("A", 2, 40960, 16, [16, 64, 128, 256], 8),          # the batch bench.py times (the metric's bs=8)
    ("S", 13, 65536, 16, [16, 64, 128, 256, 512], 1),
    ("S", 13, 65536, 16, [16, 64, 128, 256, 512], 8),    # BASELINE configs[3]: S at bs=8
    ("Kt", 20, 122880, 16, [16, 64, 128, 256], 1),
    ("Kt", 20, 122880, 16, [16, 64, 128, 256], 2),        # the per-GPU shard of BASELINE configs[4] that bench.py times
])
def test_train_step_at_benchmark_size_matches_oracle_autograd(tag, C, N, K, layers, B):
    """The path bench.py times, at its own size: _train.TrainStep's forward + dice + backward (fused pooling incl.
    d = 128, split-K, the deferred slab reducer, residual-junction BatchNorm fusion, the CSR gather backward) against the
    oracle's train-mode forward + autograd - loss, every parameter gradient (GRAD_BOUND of the arithmetic mode), train-mode
    logits (1e-3) - in the default bf16x3 arithmetic and, for config A, once more in the fp32 mode."""
    from oracle import randlanet_oracle as O
    from oracle.loss_metrics_oracle import loss_by_name
    from randlanet import _ops as ops
    from randlanet._train import TrainStep
    net, sd = _pair(C, N, K, layers, seed=23)
    net.fc_end[2].p = 0.0
    rs = np.random.RandomState(6)
    x = rs.uniform(0, 1, (B, N, 3)).astype(np.float32)
    inside = np.linalg.norm(x - 0.5, axis=-1) < 0.3
    y = np.where(inside, np.clip(1 + np.floor((C - 1) * x[..., 2]).astype(np.int64), 1, C - 1), 0).astype(np.int64)
    perm = np.random.RandomState(9).permutation(N)
    ref, ref_loss, g32, P = _oracle_step(sd, x, y, perm, layers, K)
    g64 = gyard = None
    if (tag, B) == ("A", 4):
        g64 = _oracle_step(sd, x, y, perm, layers, K, torch.float64)[2]
        gyard = {"fp32": g32, "bf16x3": _noisy_yard(sd, x, y, perm, layers, K, g64)}
    # exact fp32 products: config A, and once on the 5-layer / d = 512 shapes of S
    modes = ["bf16x3", "fp32"] if (tag == "A" or (tag, B) == ("S", 1)) else ["bf16x3"]
    default = ops.get_wide_gemm()
    try:
        for mode in modes:
            ops.set_wide_gemm(mode)
            net.load_state_dict(sd)
            net.train()
            st = TrainStep(net, B, N, loss="dice", use_graph=False)
            st.set_batch(torch.from_numpy(x).to(DEV), torch.from_numpy(y).to(DEV))
            st.perm.copy_(torch.from_numpy(perm).to(DEV))
            st._fwd_bwd()
            torch.cuda.synchronize()
            loss = float(st.out[0])
            assert abs(loss - float(ref_loss)) < 2e-5, (tag, mode, loss, float(ref_loss))
            worst = 0.0
            hip_grads = {name: st.flat.grads[name].cpu() for name, _ in net.named_parameters()}
            for name, g in hip_grads.items():
                worst = max(worst, _check_gradient((tag, mode), name, g, g32[name], GRAD_BOUND[mode]))
            if g64 is not None:
                _yardstick(f"config {tag} B={B}", mode, hip_grads, gyard[mode], g64)
            # train-mode logits (batch statistics) through the module surface, same permutation
            net.load_state_dict(sd)
            np.random.seed(0)
            state = np.random.get_state()
            perm_again = np.random.permutation(N)
            np.random.set_state(state)
            with torch.no_grad():
                logits = net(torch.from_numpy(x).to(DEV)).cpu()
            ref2 = ref if np.array_equal(perm_again, perm) else \
                O.forward({k: v.detach() for k, v in P.items()}, torch.from_numpy(x), perm_again, layer_sizes=layers,
                          n_neighbors=K, training=True, dropout_p=0.0).detach()
            lerr = float((logits - ref2.detach()).abs().max())
            print(f"[train parity] config {tag} B={B} {mode}: loss {loss:.7f} vs {float(ref_loss):.7f}, worst relative "
                  f"gradient error {worst:.2e}, train-mode logits max |diff| {lerr:.2e}")
            assert lerr < 1e-3, (tag, mode, lerr)
    finally:
        ops.set_wide_gemm(default)


# Gradient bounds per arithmetic mode.  "fp32" (exact fp32 products) is held to the 5e-3 of the parity contract.  The
# default "bf16x3" carries ~2^-16 relative error per product, and these random-weight test points are ill-conditioned: in
# the fp32 CPU oracle ITSELF a 1e-5 relative perturbation of the weights moves the logits by 4e-4 and the gradients by up
# to 3.3 % (1e-6: 4e-5 and 0.13 %; measured at the K = 32 point below).  So bf16x3 gradients are bounded by 2e-2 of each
# tensor's largest entry - the sensitivity of the function, not an arithmetic defect - while loss and logits keep 1e-5 / 1e-3
# (round 5, BatchNorm statistics as shifted sums: worst measured 1.3e-2 - config S, B = 1 - down from a 3e-2 bound).
GRAD_BOUND = {"fp32": 5e-3, "bf16x3": 2e-2}
# ... and the same statement made properly, against an fp64 evaluation of the oracle (_yardstick): the multiple of the
# yardstick's own distance from fp64 that a tensor's gradient may sit at.
# Measured on the MI355X (round 5): bf16x3 <= 2.4 on config A and <= 1.9 at the K = 32 point - the two round-4 outliers
# (BatchNorm weight gradients of level 2's residual junction at 6.6 / 10.7 x the yardstick, hence a bound of 16 then) are gone
# with the shifted statistics; fp32 mode <= 3.0 (worst: fc_end.0.conv.weight; round 4: <= 1.04).  Round 5 read the 3.0 as the
# price of these test points' RANDOM running means (oracle/init_formula.py) serving as pivots; round 6 made the pivots engine
# state that starts at ZERO - this first step runs on plain sums - and measures the same 2.95: it is the ratio of two rounding
# residues on a handful of tensors, not a pivot effect.
YARD = {"fp32": 4.0, "bf16x3": 6.0}
WIDE_NOISE = 5e-6


def test_train_step_k32_matches_oracle_autograd():
    """train.py's settings (n_points 2500, K 32, reference train.py:50-51): K != 16 takes the unfused
    gather + score GEMM + softmax-pool kernels; all gradients against the oracle's autograd, in both arithmetic modes."""
    from oracle import randlanet_oracle as O
    from oracle.loss_metrics_oracle import loss_by_name
    from randlanet import _ops as ops
    from randlanet.utils.losses import get_loss
    C, N, K, layers, B = 2, 2500, 32, [16, 64, 128, 256], 2
    net, sd = _pair(C, N, K, layers, seed=3)
    net.fc_end[2].p = 0.0
    rs = np.random.RandomState(2)
    x = rs.uniform(0, 1, (B, N, 3)).astype(np.float32)
    y = (x[..., 0] > 0.5).astype(np.int64)
    np.random.seed(11)
    perm = np.random.permutation(N)
    ref, ref_loss, g32, P = _oracle_step(sd, x, y, perm, layers, K)
    g64 = _oracle_step(sd, x, y, perm, layers, K, torch.float64)[2]
    gyard = {"fp32": g32, "bf16x3": _noisy_yard(sd, x, y, perm, layers, K, g64)}
    default = ops.get_wide_gemm()
    try:
        for mode in ("fp32", "bf16x3"):
            ops.set_wide_gemm(mode)
            net.load_state_dict(sd)
            net.zero_grad()
            net.train()
            np.random.seed(11)
            logits = net(torch.from_numpy(x).to(DEV))
            loss = get_loss("dice")(logits, torch.from_numpy(y).to(DEV))
            loss.backward()
            assert float((logits.detach().cpu() - ref.detach()).abs().max()) < 1e-3
            assert abs(float(loss.detach()) - ref_loss) < 1e-5
            worst = 0.0
            hip_grads = {name: p.grad.cpu() for name, p in net.named_parameters()}
            for name, g in hip_grads.items():
                worst = max(worst, _check_gradient((mode,), name, g, g32[name], GRAD_BOUND[mode]))
            _yardstick("K=32", mode, hip_grads, gyard[mode], g64)
            print(f"[train parity] K=32 config, {mode}: worst relative gradient error {worst:.2e} (bound {GRAD_BOUND[mode]:g})")
    finally:
        ops.set_wide_gemm(default)


@pytest.mark.parametrize("C,N,K,F,layers,B,loss_name", [
    (5, 4100, 16, 3, [8, 16, 32, 64], 3, "focal_tversky"),     # ragged: N is no multiple of 4^L; extra features; odd batch
    (3, 1029, 8, 1, [16, 32, 64], 1, "cross_entropy"),         # three levels, 8 neighbours, a single cloud
])
def test_ragged_sizes_features_and_losses_match_oracle_autograd(C, N, K, F, layers, B, loss_name):
    """Shapes the tiles do not divide (N_l = N // 4^l with remainders, rows that are no multiple of 16 / 128), extra
    point features, other losses: train-mode logits, loss and every parameter gradient against the oracle's autograd."""
    from oracle import randlanet_oracle as O
    from oracle.init_formula import formula_state_dict
    from oracle.loss_metrics_oracle import loss_by_name
    from randlanet.utils.losses import get_loss
    from randlanet.utils.modules import RandLANet, RandLANetSettings
    sd = formula_state_dict(O.state_dict_layout(C, F, layers), seed=C + N)
    net = RandLANet(RandLANetSettings(n_classes=C, n_points=N, n_features=F, n_neighbors=K, layer_sizes=list(layers)), DEV)
    net.load_state_dict(sd)
    net.fc_end[2].p = 0.0
    net.train()
    rs = np.random.RandomState(N)
    x = rs.uniform(0, 1, (B, N, 3 + F)).astype(np.float32)
    y = np.minimum((x[..., 2] * C).astype(np.int64), C - 1)
    np.random.seed(21)
    perm = np.random.permutation(N)
    np.random.seed(21)
    logits = net(torch.from_numpy(x).to(DEV))
    assert logits.shape == (B, C, N)
    loss = get_loss(loss_name)(logits, torch.from_numpy(y).to(DEV))
    loss.backward()
    P = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
         for k, v in sd.items()}
    ref = O.forward(P, torch.from_numpy(x), perm, layer_sizes=layers, n_neighbors=K, training=True, dropout_p=0.0)
    ref_loss = loss_by_name(loss_name, ref, torch.from_numpy(y))
    ref_loss.backward()
    assert float((logits.detach().cpu() - ref.detach()).abs().max()) < 1e-3
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 1e-5 * max(1.0, abs(float(ref_loss.detach())))
    for name, p in net.named_parameters():
        r = P[name].grad
        e = float((p.grad.cpu() - r).abs().max())
        assert e < 5e-3 * float(r.abs().max()) + 2e-5, (name, e, float(r.abs().max()))


def test_statistics_regrouping_does_not_move_gradients(monkeypatch):
    """Round-4 finding, round-5 regression: a mere RE-GROUPING of the BatchNorm partial sums (then: sgemm's per-lane sums after a
    transposed product) moved parameter gradients of the 1029-point ragged configuration by 0.5 - 3 % - mean / variance came out
    of sum and sum of squares by subtraction, and this configuration has nearly degenerate channels.  With the statistics as
    SHIFTED sums around the running mean (rl_gemm_desc.stats_pivot_*) the grouping must not matter: the same train step with
    the streaming GEMM on a third / a seventh of its workgroups (rl_set_sgemm_grid_div: Y bitwise equal, other lanes add other
    rows) and the wide GEMM register-staged (other per-tile grouping), exact-product mode, every gradient within 2e-4 of its
    tensor's largest entry.  The spread WITHOUT the pivot (RL_NO_BN_PIVOT) is printed next to it."""
    from oracle import randlanet_oracle as O
    from oracle.init_formula import formula_state_dict
    from randlanet import _engine as E
    from randlanet import _ops as ops
    from randlanet.utils.losses import get_loss
    from randlanet.utils.modules import RandLANet, RandLANetSettings
    C, N, K, F, layers, B = 3, 1029, 8, 1, [16, 32, 64], 1
    sd = formula_state_dict(O.state_dict_layout(C, F, layers), seed=C + N)
    # a trained network's running means sit at its batch means: start there (one step with momentum 0.99), so that the pivot
    # is what it is in a training run rather than the constructor's zeros
    rs = np.random.RandomState(N)
    x = torch.from_numpy(rs.uniform(0, 1, (B, N, 3 + F)).astype(np.float32)).to(DEV)
    y = torch.from_numpy(np.minimum((x[..., 2].cpu().numpy() * C).astype(np.int64), C - 1)).to(DEV)
    lib = ops.H.lib()

    def grads(div, staging, pivot):
        monkeypatch.setattr(E, "BN_PIVOT", pivot)
        net = RandLANet(RandLANetSettings(n_classes=C, n_points=N, n_features=F, n_neighbors=K, layer_sizes=list(layers)), DEV)
        net.load_state_dict(sd)
        net.fc_end[2].p = 0.0
        net.train()
        out = None
        for it in range(2):            # the first step moves the running means onto the batch; the second is the one compared
            net.zero_grad()
            assert lib.rl_set_sgemm_grid_div(div if it else 1) == 0
            ops.set_wgemm_staging(staging if it else "dma")
            np.random.seed(21)
            logits = net(x)
            get_loss("cross_entropy")(logits, y).backward()
            out = {n: p.grad.detach().cpu().clone() for n, p in net.named_parameters()}
        return out

    default = ops.get_wide_gemm()
    try:
        ops.set_wide_gemm("fp32")
        for pivot in (True, False):
            base = grads(1, "dma", pivot)
            worst = (0.0, "")
            for div, staging in ((3, "dma"), (7, "dma"), (1, "registers"), (5, "registers")):
                g = grads(div, staging, pivot)
                for name, r in base.items():
                    d = float((g[name] - r).abs().max())
                    if name == "fc_start.bias" or (name.endswith("conv.bias") and name != "fc_end.3.conv.bias"):
                        # a bias in front of a BatchNorm: its TRUE gradient is 0 and what is computed is the rounding residue of
                        # a cancelling sum (1e-9 ... 1e-7 here, another value for every grouping) - held absolutely, not relative
                        # to itself (round 6: the pivots moved onto the batch mean and one residue went from 4e-8 to 1.04e-7)
                        assert d < 1e-6, (name, d)
                        continue
                    e = max(0.0, d - 1e-7) / (float(r.abs().max()) + 1e-12)
                    worst = max(worst, (e, name))
            print(f"[regrouping] shifted sums {'on ' if pivot else 'off'}: worst relative gradient difference {worst[0]:.2e} ({worst[1]})")
            if pivot:
                assert worst[0] < 2e-4, worst
    finally:
        ops.set_wide_gemm(default)
        ops.set_wgemm_staging("dma")
        lib.rl_set_sgemm_grid_div(1)
