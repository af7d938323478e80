"""Forward / backward schedule of the RandLA-Net hot path over the HIP kernels.

This is the host side of RandLANet.forward (reference randlanet/utils/modules.py:542-611) and
of its autograd backward, written as an explicit launch schedule: every step below is one call
into librandla_hip.so (include/rl_randlanet.h) on the current HIP stream, with no host
synchronisation, so a whole training step can be captured into one hipGraph (see bench.py).

Layout: activations are point-major / channel-last (B*n rows of C floats) instead of the
reference's (B,C,N,K); parameters stay in the reference's state_dict layout and are read in
place through strides, so nothing is converted at the boundary except the logits
((B,C,N), original point order).

Representation: the output of a SharedMLP is kept RAW (pre-BatchNorm) together with the
per-channel (scale, shift) of its BatchNorm and its activation (`_ops.Lazy`); consumers apply
them while loading.  BatchNorm therefore costs one tiny finalize kernel, not a pass over memory.

Random sampling: the permutation (modules.py:571-573) is applied once to the input rows; all
levels are prefixes of the permuted order (modules.py:587-598), read through batch strides;
fc_end runs in permuted order (it is per-point, and BatchNorm statistics are order-invariant)
and only the logits are un-permuted (modules.py:608).
"""
from typing import Dict, List, Optional

import torch

from . import _hip as H
from . import _ops as ops
from ._ops import Lazy, Rpe

BN_EPS = 1e-6       # modules.py:87, :497
BN_MOMENTUM = 0.99
# the CSR transposes of the neighbour graphs (backward only) on a second stream beside the forward: ONE fork and ONE join
# in the replayed graph - measured 7.96 -> 8.12 ms per step (a cross-stream edge costs more than the 0.3 ms it could hide), so OFF
CSR_SIDE_STREAM = bool(int(__import__("os").environ.get("RL_CSR_SIDE_STREAM", "0")))
FOLD_BIAS = not bool(int(__import__("os").environ.get("RL_NO_FOLD_BIAS", "0")))     # diagnostics: keep the bias in the GEMM
# BatchNorm batch statistics as SHIFTED sums around the running mean (round 5): var = E[(y-c)^2] - E[y-c]^2 keeps the variance
# of a channel whose spread is tiny against its mean, which E[y^2] - E[y]^2 on fp32 partial sums loses.  A/B: RL_NO_BN_PIVOT=1
BN_PIVOT = not bool(int(__import__("os").environ.get("RL_NO_BN_PIVOT", "0")))
# A/B: pivot on the running mean (round 5) instead of the engine's own pivot vectors (the previous batch's mean, round 6)
BN_PIVOT_RUNNING = bool(int(__import__("os").environ.get("RL_BN_PIVOT_RUNNING", "0")))
FC_START_GENERIC = bool(int(__import__("os").environ.get("RL_FC_START_GENERIC", "0")))      # A/B: fc_start on the unpadded rows
# clouds below this size keep the permutation as drawn (a level-0 table of < 128 KB sits in L2 / L1 whatever the order)
BAND_SORT_MIN_POINTS = 4096


class _Tape(list):
    """The forward's records; each remembers the encoder level it was appended under (ops.LEVEL), so that the backward's
    launches carry the same tag in the kernel timer."""

    def __init__(self):
        super().__init__()
        self.levels: List[int] = []

    def append(self, rec) -> None:
        super().append(rec)
        self.levels.append(ops.LEVEL)

    def clear(self) -> None:
        super().clear()
        self.levels.clear()


class Context:
    """What one forward leaves behind for its backward."""

    def __init__(self):
        self.tape: "_Tape" = _Tape()
        self.grads: Dict[int, list] = {}     # id(raw tensor) -> [gradient tensor, initialised]
        self.keep: List[torch.Tensor] = []
        self.training = False
        self.B = self.N = 0
        self.perm: Optional[torch.Tensor] = None
        self.logits_perm: Optional[Lazy] = None


class Prep:
    """The coordinate-only part of a forward (Engine.prepare)."""

    def __init__(self):
        self.B = self.N = 0
        self.training = False
        self.inp_p = self.xyz = self.xyz4 = None
        self.searches = self.csrs = None
        self.csr_ready = None


class Engine:
    def __init__(self, layer_sizes, n_neighbors: int, decimation: int, n_classes: int, n_features: int,
                 params: Dict[str, torch.Tensor], buffers: Dict[str, torch.Tensor]):
        self.layers = list(layer_sizes)
        self.K = int(n_neighbors)
        self.dec = int(decimation)
        self.C = int(n_classes)
        self.F = int(n_features)
        self.P = params      # reference state_dict names -> tensors (parameters)
        self.Bf = buffers    # running_mean / running_var / num_batches_tracked
        # pivots of the shifted batch statistics, one vector per BatchNorm layer: the PREVIOUS training batch's mean of the
        # layer's output, left there by rl_bn_finalize (round 6; round 5 pivoted on the running mean, which after a
        # load_state_dict can sit ten standard deviations off the data: var = Q/n - (S/n)^2 then loses two digits).  Zero for
        # a fresh or freshly loaded model (reset_pivots): the first step runs on plain sums.  Engine state, not module state:
        # the state_dict keeps the reference's 262 / 320 entries.  Allocated HERE (a captured forward must not allocate them)
        self.Pv: Dict[str, torch.Tensor] = {k[:-len(".running_mean")]: torch.zeros_like(v) for k, v in buffers.items()
                                             if k.endswith(".running_mean")}
        self._side: Optional[torch.cuda.Stream] = None   # weight gradients run beside the dgrad chain
        # eval mode: the BatchNorm layers of a forward (name -> (C, folded conv bias)), see _fold - one table per SIGNATURE of the
        # forward (which levels run the virtual rpe branch: it folds mlp_rpe1/2's BatchNorm without the conv bias, the stored
        # branch with it, and which one a level takes depends on the batch's shape)
        self._eval_specs: Dict[tuple, dict] = {}
        self._eval_spec_build = {}
        # Dropout(fc_end): Philox mask keyed by (seed, pass counter); the counter lives on the device so that captured
        # graphs draw a new mask per replay.  The seed comes from torch's seed WITHOUT consuming its random stream.
        # data-parallel equivalence mode: BatchNorm statistics of the global batch (ops.SyncGroup)
        self.sync: Optional[ops.SyncGroup] = None
        self._drop_seed = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
        # allocated (and zeroed) HERE, eagerly: a lazy torch.zeros inside a captured forward would become a memset node
        # of the graph and reset the counter - the same mask - on every replay
        dev = next(iter(params.values())).device
        self._drop_counter: Optional[torch.Tensor] = torch.zeros(1, dtype=torch.int64, device=dev) if dev.type == "cuda" else None
        # data-parallel runs: `drop_stream` decorrelates the masks of ranks that train on DIFFERENT batches (plain DDP:
        # the rank); `sync.first_row(N)` places a shard inside the whole batch's mask in the equivalence mode
        self.drop_stream = 0

    # ------------------------------------------------------------------------ forward pieces
    def _w2(self, name: str) -> torch.Tensor:
        w = self.P[name]
        return w.view(w.shape[0], w.shape[1])

    def _pivot(self, ctx: Context, bn_name: str):
        """The pivot of a layer's shifted batch statistics: the previous batch's mean (training only), see self.Pv."""
        if not (ctx.training and BN_PIVOT):
            return None
        return self.Bf[f"{bn_name}.running_mean"] if BN_PIVOT_RUNNING else self.Pv[bn_name]

    def reset_pivots(self) -> None:
        """After the weights were replaced (load_state_dict): the pivots of the old weights' activations mean nothing."""
        for t in self.Pv.values():
            t.zero_()

    def _fold(self, ctx: Context, bn_name: str, stats, rows: int, C: int, folded_bias=None, nslots=None):
        """The BatchNorm fold of one layer (rl_bn_finalize).  Eval mode: the fold depends on nothing the forward computes (running
        statistics, gamma, beta), so the folds of ALL layers are issued in front of the network as grouped launches
        (_eval_folds: 2 launches instead of 24 dependent ones sprinkled over the chain - 6 % of an eval forward); the first
        eval forward of an engine runs them one by one and records which layers there are."""
        if not ctx.training:
            pre = getattr(ctx, "eval_folds", None)
            if pre is not None and bn_name in pre:
                scale, shift, spec_C, spec_fb = pre[bn_name]
                if spec_C == C and spec_fb is folded_bias:          # (a fold made for another path is not this layer's fold)
                    return scale, shift, None, None
            self._eval_spec_build[bn_name] = (C, folded_bias)
        nbt = self.Bf.get(f"{bn_name}.num_batches_tracked")
        return ops.bn_finalize(
            stats, rows, 128, C, self.P[f"{bn_name}.weight"], self.P[f"{bn_name}.bias"],
            self.Bf[f"{bn_name}.running_mean"], self.Bf[f"{bn_name}.running_var"],
            nbt if ctx.training else None, BN_MOMENTUM, BN_EPS, ctx.training, sync=self.sync, folded_bias=folded_bias,
            nslots=nslots, defer=getattr(ctx, "bn_defer", None), pivoted=ctx.training and BN_PIVOT,
            pivot=None if BN_PIVOT_RUNNING else self._pivot(ctx, bn_name))

    def _eval_folds(self, spec: dict):
        """(scale, shift) of every BatchNorm layer from the running statistics, as grouped launches."""
        lst, out = [], {}
        for bn_name, (C, fb) in spec.items():
            scale, shift, _, _ = ops.bn_finalize(
                None, 1, 128, C, self.P[f"{bn_name}.weight"], self.P[f"{bn_name}.bias"], self.Bf[f"{bn_name}.running_mean"],
                self.Bf[f"{bn_name}.running_var"], None, BN_MOMENTUM, BN_EPS, False, folded_bias=fb, nslots=1, defer=lst)
            out[bn_name] = (scale, shift, C, fb)
        ops.bn_finalize_flush(lst)
        return out

    def _bn(self, ctx: Context, out: Lazy, stats, bn_name: str, act: int, slope: float, folded_bias=None, nslots=None):
        scale, shift, mean, invstd = self._fold(ctx, bn_name, stats, out.rows, out.C, folded_bias, nslots)
        out.scale, out.shift, out.mean, out.invstd = scale, shift, mean, invstd
        out.act, out.slope, out.bn = act, slope, bn_name

    def _linear(self, ctx: Context, a, wname: str, bname: Optional[str], n_out: int, *, transposed=False,
                bn: Optional[str] = None, act: int = H.ACT_NONE, slope: float = 0.0, a_grad: bool = True) -> Lazy:
        """Y = A'.W + b  (+ lazy BatchNorm/activation).  Records the layer for backward."""
        W = self._w2(wname)
        K = 10 if isinstance(a, Rpe) else a.C
        ks, ns = ops.weight_strides(W, transposed, K, n_out)
        stats = ops.new_stats(W.device, n_out) if (bn and ctx.training) else None
        # a bias in front of a BatchNorm cancels in (y - mean): the GEMM epilogue leaves it out (a bias costs a wide GEMM
        # launch +18 %) and the BatchNorm fold accounts for it where it shows - the running mean (rl_bn_finalize)
        fold = FOLD_BIAS and bn is not None and bname is not None
        piv = self._pivot(ctx, bn) if stats is not None else None
        Y = ops.gemm(a, W, ks, ns, n_out, self.P[bname] if (bname and not fold) else None, stats=stats,
                     wsplit=getattr(ctx, "wsplit", None), pivot=(piv, self.P[bname] if fold else None) if piv is not None else None)
        rpb = a.n * a.K if isinstance(a, Rpe) else a.n
        out = Lazy(Y, a.B, rpb, rpb, n_out)
        if bn:
            self._bn(ctx, out, stats, bn, act, slope, folded_bias=self.P[bname] if fold else None,
                     nslots=ops.gemm_stat_slots(out.rows, n_out, K) if stats is not None else None)
        else:
            assert act == H.ACT_NONE
        ctx.tape.append(("linear", a, out, wname, bname, ks, ns, a_grad))
        return out

    def _mlp_pair(self, ctx, a: Lazy, first: tuple, second: tuple):
        """Two SharedMLPs over the same input (mlp1 and shortcut of an encoder level, modules.py:314, 325) as ONE wide-GEMM launch
        where rl_gemm_pair takes them; first / second: (name, n_out, act, slope).  The tape gets the two "linear" records it would
        get from two _mlp calls (the backward is theirs)."""
        specs, metas = [], []
        for name, n_out, act, slope in (first, second):
            wname, bname, bn = f"{name}.conv.weight", f"{name}.conv.bias", f"{name}.batch_norm"
            W = self._w2(wname)
            ks, ns = ops.weight_strides(W, False, a.C, n_out)
            stats = ops.new_stats(W.device, n_out) if ctx.training else None
            piv = self._pivot(ctx, bn) if stats is not None else None
            if not FOLD_BIAS:
                return self._mlp(ctx, a, first[0], first[1], first[2], first[3]), self._mlp(ctx, a, second[0], second[1], second[2], second[3])
            specs.append((W, ks, ns, n_out, stats, (piv, self.P[bname]) if piv is not None else None))
            metas.append((wname, bname, bn, n_out, act, slope, ks, ns, stats))
        res = ops.gemm_pair(a, specs[0], specs[1], getattr(ctx, "wsplit", None))
        if res is None:
            return self._mlp(ctx, a, first[0], first[1], first[2], first[3]), self._mlp(ctx, a, second[0], second[1], second[2], second[3])
        outs = []
        for Y, (wname, bname, bn, n_out, act, slope, ks, ns, stats) in zip(res, metas):
            out = Lazy(Y, a.B, a.n, a.n, n_out)
            self._bn(ctx, out, stats, bn, act, slope, folded_bias=self.P[bname],
                     nslots=H.row_blocks(out.rows, 128) if stats is not None else None)
            ctx.tape.append(("linear", a, out, wname, bname, ks, ns, True))
            outs.append(out)
        return outs[0], outs[1]

    def _mlp(self, ctx, a, name: str, n_out: int, act: int = H.ACT_NONE, slope: float = 0.0, *, transposed=False,
             bn=True, a_grad=True) -> Lazy:
        """SharedMLP (modules.py:60-104)."""
        return self._linear(ctx, a, f"{name}.conv.weight", f"{name}.conv.bias", n_out, transposed=transposed,
                            bn=f"{name}.batch_norm" if bn else None, act=act, slope=slope, a_grad=a_grad)

    def _pool(self, ctx, name: str, u, g: Lazy, idx: torch.Tensor, csr, n: int, d: int, n_out: int, stage: int = 0,
              next_stats: bool = False) -> Lazy:
        """PointFeatureAugmentation + AttentivePooling (modules.py:213-221, 246-253)."""
        B, K, h = u.B, self.K, d // 2
        rows = B * n * K
        Ws = self.P[f"{name}.score_fn.0.weight"]
        if ops.pool_supported(d, K):
            # narrow levels: one fused kernel, nothing of size rows x d touches HBM
            if next_stats:
                res, st2, ns2 = ops.pool_fwd(u, g, idx, Ws, n, d, stage, next_stats=True)
                ctx.next_stats = (st2, ns2)
            else:
                res = ops.pool_fwd(u, g, idx, Ws, n, d, stage)
            pooled = ops.plain(res, B, n)
            ctx.tape.append(("pool_fused", name, u, g, csr, idx, pooled, n, d, stage))
            return self._mlp(ctx, pooled, f"{name}.mlp", n_out, H.ACT_RELU)
        X = torch.empty((rows, d), dtype=torch.float32, device=u.raw.device)
        ops.copy_rows_pair(((u.raw, (0, h), n * K, X, (0, h), rows, n * K), dict(lazy=u)),
                           ((g.raw, (0, h), g.bstride, X, (h, h), rows, n * K), dict(index=idx, lazy=g)))
        S = ops.gemm(ops.plain(X, B, n * K), Ws, 1, d, d, None, wsplit=getattr(ctx, "wsplit", None))
        Pt = ops.attpool_fwd(X, S, B * n, K)
        pooled = ops.plain(Pt, B, n)
        ctx.tape.append(("pool", name, u, g, csr, X, S, pooled, n, d))
        return self._mlp(ctx, pooled, f"{name}.mlp", n_out, H.ACT_RELU)

    def _virtual_bn(self, ctx: Context, vr, stage: int, bn_name: str, stats=None, nslots: int = 1) -> Lazy:
        """BatchNorm of a virtual rpe stage: statistics from rl_rpe_stats / the previous pooling kernel (training) or the
        running ones."""
        if ctx.training and stats is None:
            stats, nslots = ops.rpe_stats(vr, stage)
        scale, shift, mean, invstd = self._fold(ctx, bn_name, stats, vr.rows, vr.h, None, nslots)
        rec = Lazy(vr.d2, vr.B, vr.n * 16, vr.n * 16, vr.h, scale, shift, H.ACT_RELU, 0.0, mean, invstd, bn_name)
        return rec

    def _lfa(self, ctx, l: int, xin: Lazy, xyz: torch.Tensor, n: int, d: int, idx, d2, csr=None) -> Lazy:
        """LocalFeatureAggregation (modules.py:298-325); idx / d2 = its K nearest neighbours."""
        e = f"encoder.{l}"
        B, K, h = xin.B, self.K, d // 2
        ctx.keep += [idx, d2]
        # BatchNorm folds wanted at the same moment go out as ONE launch: (mlp1, shortcut, mlp_rpe1), then (pool1.mlp, mlp_rpe2).
        # ctx.bn_defer collects them; every flush sits in front of the first kernel that reads one of the (scale, shift) pairs
        ctx.bn_defer = [] if self.sync is None else None
        f0, sc = self._mlp_pair(ctx, xin, (f"{e}.mlp1", h, H.ACT_LRELU, 0.2), (f"{e}.shortcut", 2 * d, H.ACT_NONE, 0.0))
        if ops.virtual_rpe_supported(d, K, B * n, n):
            # the outputs of mlp_rpe1 / mlp_rpe2 are never stored: their consumers recompute them from the coordinates
            vr = ops.VirtualRpe(ctx.xyz4 if getattr(ctx, "xyz4", None) is not None else xyz, idx, d2, B, n, h, self.P[f"{e}.mlp_rpe1.conv.weight"], self.P[f"{e}.mlp_rpe1.conv.bias"],
                                self.P[f"{e}.mlp_rpe2.conv.weight"], self.P[f"{e}.mlp_rpe2.conv.bias"])
            vr.piv1, vr.piv2 = self._pivot(ctx, f"{e}.mlp_rpe1.batch_norm"), self._pivot(ctx, f"{e}.mlp_rpe2.batch_norm")
            vr.bn1 = self._virtual_bn(ctx, vr, 1, f"{e}.mlp_rpe1.batch_norm")
            ops.bn_finalize_flush(ctx.bn_defer)
            # in training pool1's kernel also leaves the batch statistics of mlp_rpe2's raw output (it has the tile)
            q1 = self._pool(ctx, f"{e}.pool1", vr, f0, idx, csr, n, d, h, stage=1, next_stats=ctx.training)
            vr.bn2 = self._virtual_bn(ctx, vr, 2, f"{e}.mlp_rpe2.batch_norm", *getattr(ctx, "next_stats", (None, 1)))
            ctx.next_stats = (None, 1)
            ops.bn_finalize_flush(ctx.bn_defer)
            ctx.bn_defer = None
            q2 = self._pool(ctx, f"{e}.pool2", vr, q1, idx, csr, n, d, d, stage=2)
        else:
            rpe = Rpe(xyz, idx, d2, B, n, K)
            if h <= 128 and not ops.NO_RPE_TENSOR:
                # read twice (forward + weight gradient): cheaper as a 48-byte row than re-gathered in both kernels, and
                # a plain 12-float row lets mlp_rpe1 run on the streaming GEMM / weight-gradient kernels (up to 128 columns)
                rpe = ops.rpe_build(rpe)
            u1 = self._mlp(ctx, rpe, f"{e}.mlp_rpe1", h, H.ACT_RELU, a_grad=False)
            ops.bn_finalize_flush(ctx.bn_defer)
            q1 = self._pool(ctx, f"{e}.pool1", u1, f0, idx, csr, n, d, h)
            u2 = self._mlp(ctx, u1, f"{e}.mlp_rpe2", h, H.ACT_RELU)
            ops.bn_finalize_flush(ctx.bn_defer)
            ctx.bn_defer = None
            q2 = self._pool(ctx, f"{e}.pool2", u2, q1, idx, csr, n, d, d)
        m2 = self._mlp(ctx, q2, f"{e}.mlp2", 2 * d)
        O = ops.plain(ops.add_act_fwd(m2, sc, 0.01), B, n)
        ctx.tape.append(("add_act", m2, sc, O))
        return O

    def _wide_weight_uses(self, training: bool):
        """(W, w_ks, w_ns, K, N) of every product the LDS-tiled wide GEMM will run this pass: forward orientation, and the
        dgrad orientation (strides swapped, K and N exchanged) when training."""
        uses = []
        for name, W in self.P.items():
            if not name.endswith(("conv.weight", "score_fn.0.weight")) or W.dim() < 2:
                continue
            a, b = int(W.shape[0]), int(W.shape[1])
            if max(a, b) <= 64:
                # mlp1 of a level whose shortcut is wide rides along with it (ops.gemm_pair): forward planes of the narrow weight too
                if name.endswith(".mlp1.conv.weight") and b % 32 == 0 and b > 64 and not ops.NO_GEMM_PAIR:
                    sc = self.P.get(name.replace(".mlp1.", ".shortcut."))
                    if sc is not None and int(sc.shape[0]) > 64:
                        uses.append((W.view(a, b), 1, b, b, a, True))
                continue
            transposed = name.startswith("decoder.")                 # ConvTranspose2d weights are (in, out)
            K, N = (a, b) if transposed else (b, a)
            ks, ns = (N, 1) if transposed else (1, K)
            W2 = W.view(a, b)
            uses.append((W2, ks, ns, K, N))
            if training:
                uses.append((W2, ns, ks, N, K))                      # dA = dY . W^T
        return uses

    # ------------------------------------------------------------------------------ forward
    def min_points(self) -> int:
        L = len(self.layers)
        return max(self.K * self.dec ** (L - 1), 2 * self.dec ** L)       # modules.py:488-491

    def prepare(self, inp: torch.Tensor, perm: torch.Tensor, training: bool) -> "Prep":
        """Everything of a forward that depends on the INPUT ROWS AND THE PERMUTATION ALONE - no weight, no activation: the
        permuted rows (modules.py:571-573), every neighbour search of the pass (the K-NN of the L encoder levels, modules.py:309,
        and the 1-NN of the L decoder steps, modules.py:358, as one batch), the padded coordinates, and - training - the transpose
        of every neighbour graph.  A caller may run this ahead of / beside the network (`_train.TrainStep`: as its own graph on a
        second stream, under the previous step's kernels) and hand the result to forward(prep=...)."""
        B, N, cin = inp.shape
        assert cin == 3 + self.F and inp.dtype == torch.float32 and inp.is_cuda and inp.is_contiguous()
        assert perm.dtype == torch.int64 and perm.numel() == N and perm.is_cuda
        assert N >= self.min_points()
        dev = inp.device
        L, dec = len(self.layers), self.dec
        prep = Prep()
        prep.B, prep.N, prep.training = B, N, training
        # random permutation of the rows (modules.py:571-573), once, on the input.  The reference sub-samples by PREFIXES of the
        # permuted order (modules.py:587-598), so only the band [N / dec^(l+1), N / dec^l) a point falls in matters; inside a band
        # the points are put in cell order, cloud by cloud (ops.band_sort: same sampled sets, neighbour gathers that follow space)
        prep.perm = perm
        edges = [0] + [N // dec ** l for l in range(L, -1, -1)]
        if (not ops.NO_BAND_SORT and N >= BAND_SORT_MIN_POINTS and len(edges) - 1 <= ops.BAND_SORT_MAX_BANDS
                and all(a < b for a, b in zip(edges[:-1], edges[1:]))):
            prep.perm = ops.band_sort(inp, perm, edges)
        inp_p = torch.empty((B * N, cin), dtype=torch.float32, device=dev)
        # coordinates padded to 16 bytes for the kernels that gather them per neighbour (virtual rpe branch) and for fc_start
        want4 = any(ops.virtual_rpe_supported(d, self.K, B * (N // dec ** l), N // dec ** l) for l, d in enumerate(self.layers))
        xyz4 = torch.empty((B, N, 4), dtype=torch.float32, device=dev) if want4 else None
        gather = dict(index=prep.perm.view(-1), index_shared=prep.perm.dim() == 1)
        if want4 and cin == 3:
            # (both from the permutation gather, one launch: the padded copy used to be a launch of its own behind the searches)
            ops.copy_rows_pair(((inp.view(B * N, 3), (0, 3), N, inp_p, (0, 3), B * N, N), gather),
                               ((inp.view(B * N, 3), (0, 3), N, xyz4.view(B * N, 4), (0, 3), B * N, N), gather))
        else:
            ops.copy_rows(inp.view(B * N, cin), (0, cin), N, inp_p, (0, cin), B * N, N, **gather)
        if cin == 3:
            xyz = inp_p.view(B, N, 3)
        else:
            xyz = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
            ops.copy_rows(inp_p, (0, 3), N, xyz.view(B * N, 3), (0, 3), B * N, N)
        prep.inp_p, prep.xyz = inp_p, xyz

        # every neighbour search of this forward depends on the coordinates only: the K-NN of the L encoder
        # levels (modules.py:309) and the 1-NN of the L decoder steps (modules.py:358) go out as one batch
        tasks, ratio = [], 1
        for _ in self.layers:
            tasks.append((N // ratio, N // ratio, self.K))
            ratio *= dec
        for _ in self.layers:
            tasks.append((N // ratio, dec * N // ratio, 1))
            ratio //= dec
        searches = ops.knn_multi(xyz, tasks)
        if want4 and cin != 3:
            ops.copy_rows(xyz.view(B * N, 3), (0, 3), N, xyz4.view(B * N, 4), (0, 3), B * N, N)
        prep.xyz4, prep.searches = xyz4, searches
        # training: the transpose of every neighbour graph ("who gathered from me"), so that the gathers' backward sums
        # each destination row in a fixed order (bitwise reproducible steps; torch's scatter_add_ has no defined order)
        csrs = [None] * (2 * L)
        prep.csr_ready = None
        if training and CSR_SIDE_STREAM:
            # only the backward needs the transposes, and they depend on the neighbour indices alone: built on a second
            # stream beside the forward (ONE fork here, ONE join at the top of backward - two graph edges, not two per layer)
            main = torch.cuda.current_stream(dev)
            if self._side is None:
                self._side = torch.cuda.Stream(dev)
            fork = torch.cuda.Event()
            fork.record(main)
            self._side.wait_event(fork)
            with torch.cuda.stream(self._side):
                csrs = ops.csr_build([(searches[i][0], tasks[i][0]) for i in range(2 * L)])
                prep.csr_ready = torch.cuda.Event()
                prep.csr_ready.record(self._side)
        elif training:
            csrs = ops.csr_build([(searches[i][0], tasks[i][0]) for i in range(2 * L)])
        prep.csrs = csrs
        return prep

    def forward(self, inp: torch.Tensor, perm: torch.Tensor, training: bool, dropout_p: float = 0.5,
                keep_mask: Optional[torch.Tensor] = None, logits_out: Optional[torch.Tensor] = None,
                prep: Optional["Prep"] = None, head: Optional[ops.Head] = None):
        """inp (B,N,3+F) fp32 on the device, perm (N,) int64 on the device -> logits (B,C,N), ctx.
        keep_mask (B*N, 32) uint8: an explicit Dropout mask (parity tests); by default the mask is generated in the kernel.
        prep: the result of prepare(inp, perm, training) when the caller has run it already (same inp / perm contents).
        head (training): the labels / loss of the step - Dropout, fc_end.3, the un-permute and the loss then run as ONE kernel
        (rl_head_fwd) that fills head.out; no logits are stored and (None, ctx) is returned; backward(ctx, None, grads) starts
        from rl_head_bwd.  Without it (or where rl_head_supported says no) the layers run one by one and the logits come back."""
        B, N, cin = inp.shape
        assert cin == 3 + self.F and inp.dtype == torch.float32 and inp.is_cuda and inp.is_contiguous()
        assert perm.dtype == torch.int64 and perm.numel() == N and perm.is_cuda
        assert N >= self.min_points()
        dev = inp.device
        ctx = Context()
        ctx.training, ctx.B, ctx.N, ctx.perm = training, B, N, perm
        ctx.pending_ok = not (ops.SIDE_STREAM_WGRAD or ops.NO_DEFERRED_WGRAD)      # (the fused head queues its slabs on ctx.pending)
        L, dec = len(self.layers), self.dec
        ctx.eval_folds = None
        ctx.eval_sig = tuple(ops.virtual_rpe_supported(d, self.K, B * (N // dec ** l), N // dec ** l) for l, d in enumerate(self.layers))
        if not training:
            self._eval_spec_build = {}
            spec = self._eval_specs.get(ctx.eval_sig)
            if spec is not None and self.sync is None and not ops.NO_BN_BATCH:
                ctx.eval_folds = self._eval_folds(spec)

        # wide layers: bf16 head / tail planes of their weights, both orientations, in one launch (the weights change every step)
        ctx.wsplit = ops.split_weights(self._wide_weight_uses(training))
        if prep is None:
            prep = self.prepare(inp, perm, training)
        perm = prep.perm             # (B, N) when the bands were put in cell order: what the un-permute / the head index with
        ctx.perm = perm
        assert prep.B == B and prep.N == N and prep.training >= training
        inp_p, xyz, searches, csrs = prep.inp_p, prep.xyz, prep.searches, prep.csrs
        ctx.keep += [inp_p, xyz]
        if prep.xyz4 is not None:
            ctx.keep.append(prep.xyz4)
        ctx.xyz4 = prep.xyz4
        ctx.csr_ready = prep.csr_ready
        prep.csr_ready = None

        # fc_start + bn_start (modules.py:565-566)
        # (coordinates only: the rows padded to 16 bytes - made for the virtual rpe branch - let fc_start's K = 3 product and its weight
        # gradient run on the streaming kernels, 16-byte loads, instead of the generic gemm_kernel: 18 -> 6 us at 8 clouds)
        a0 = ops.plain(inp_p, B, N)
        if cin == 3 and prep.xyz4 is not None and not FC_START_GENERIC:
            a0 = Lazy(prep.xyz4.view(B * N, 4), B, N, N, 3)
        x = self._linear(ctx, a0, "fc_start.weight", "fc_start.bias", 8, bn="bn_start.0",
                         act=H.ACT_LRELU, slope=0.2, a_grad=False)
        # encoder (modules.py:582-589)
        skips: List[Lazy] = []
        ratio = 1
        for l, d in enumerate(self.layers):
            n_l = N // ratio
            ops.LEVEL = l
            x = self._lfa(ctx, l, x.prefix(n_l), xyz, n_l, d, *searches[l], csr=csrs[l])
            ops.LEVEL = -1
            skips.append(x)
            ratio *= dec
        x = self._mlp(ctx, x.prefix(N // ratio), "mlp", x.C, H.ACT_RELU)        # modules.py:591
        # decoder (modules.py:594-605)
        for j in range(L):
            n_c, n_f = N // ratio, dec * N // ratio
            nn = searches[L + j][0]
            assert nn.shape == (B, n_f, 1)
            skip = skips.pop()
            assert skip.n == n_f and x.n == n_c
            cat = torch.empty((B * n_f, x.C + skip.C), dtype=torch.float32, device=dev)
            ops.copy_rows_pair(((x.raw, (0, x.C), x.bstride, cat, (0, x.C), B * n_f, n_f), dict(index=nn, lazy=x)),
                               ((skip.raw, (0, skip.C), skip.bstride, cat, (x.C, skip.C), B * n_f, n_f), {}))
            catl = ops.plain(cat, B, n_f)
            catl.concat = (x, skip)            # (backward: its gradient may leave the dgrad GEMM in two pieces, see _bwd_linear)
            ctx.tape.append(("interp_concat", x, skip, csrs[L + j], catl))
            n_out = 8 if j == L - 1 else 2 * self.layers[L - 2 - j]
            x = self._mlp(ctx, catl, f"decoder.{j}", n_out, H.ACT_RELU, transposed=True)
            ratio //= dec
        # fc_end in permuted order; the logits are un-permuted at the very end (modules.py:608-611)
        x = self._mlp(ctx, x, "fc_end.0", 64, H.ACT_RELU)
        x = self._mlp(ctx, x, "fc_end.1", 32, H.ACT_RELU)
        if (head is not None and training and self.sync is None and keep_mask is None and ctx.pending_ok
                and ops.head_supported(x, self.C)):
            key, seed, first_row = None, 0, 0
            if dropout_p > 0.0:
                if self._drop_counter is None or self._drop_counter.device != dev:
                    if torch.cuda.is_current_stream_capturing():
                        raise H.HipKernelError("the Dropout counter must exist before a forward is captured")
                    self._drop_counter = torch.zeros(1, dtype=torch.int64, device=dev)
                key = ops.dropout_tick(self._drop_counter)
                seed = (self._drop_seed + 0x9E3779B97F4A7C15 * self.drop_stream) & 0x7FFFFFFFFFFFFFFF
            drop = (key, seed, dropout_p, first_row)
            ops.head_fwd(x, self._w2("fc_end.3.conv.weight"), self.P["fc_end.3.conv.bias"], perm, head, drop)
            ctx.tape.append(("head", x, head, drop))
            ctx.logits_perm = None
            return None, ctx
        if training and dropout_p > 0.0:
            if keep_mask is None:
                if self._drop_counter is None or self._drop_counter.device != dev:
                    if torch.cuda.is_current_stream_capturing():
                        raise H.HipKernelError("the Dropout counter must exist before a forward is captured")
                    self._drop_counter = torch.zeros(1, dtype=torch.int64, device=dev)
                key = ops.dropout_tick(self._drop_counter)
                seed = (self._drop_seed + 0x9E3779B97F4A7C15 * self.drop_stream) & 0x7FFFFFFFFFFFFFFF
                first_row = self.sync.cloud_offset * N if self.sync is not None else 0
                dropped = ops.plain(ops.dropout_fwd(x, key, seed, dropout_p, first_row), B, N)
                ctx.tape.append(("dropout_philox", x, dropped, key, dropout_p, seed, first_row))
            else:
                t = torch.empty((B * N, 32), dtype=torch.float32, device=dev)
                ops.copy_rows(x.raw, (0, 32), N, t, (0, 32), B * N, N, lazy=x)
                ops.scale_mask(t, keep_mask, 1.0 / (1.0 - dropout_p))
                dropped = ops.plain(t, B, N)
                ctx.tape.append(("dropout", x, dropped, keep_mask, 1.0 / (1.0 - dropout_p)))
            x = dropped
        lp = self._mlp(ctx, x, "fc_end.3", self.C, bn=False)
        ctx.logits_perm = lp
        if not training and ctx.eval_folds is None and self._eval_spec_build:
            self._eval_specs[ctx.eval_sig] = dict(self._eval_spec_build)        # (the next eval forward of this signature folds all of them up front)
        logits = ops.logits_unpermute(lp.raw, perm, B, N, out=logits_out)
        return logits, ctx

    # ----------------------------------------------------------------------------- backward
    @staticmethod
    def _gbuf(ctx: Context, lz: Lazy):
        ent = ctx.grads.get(id(lz.raw))
        if ent is None:
            ent = [torch.empty_like(lz.raw), False]
            ctx.grads[id(lz.raw)] = ent
        return ent

    def backward(self, ctx: Context, dlogits: torch.Tensor, grads: Dict[str, torch.Tensor]) -> None:
        """dlogits (B,C,N) -> fills grads[name] for every parameter (reference names/layouts)."""
        if not ctx.training:
            raise H.HipKernelError("backward is implemented for training-mode forwards (batch statistics)")
        B, N = ctx.B, ctx.N
        fused_head = bool(ctx.tape) and ctx.tape[-1][0] == "head"
        if not fused_head:
            assert dlogits.shape == (B, self.C, N) and dlogits.is_cuda
            dlogits = dlogits.contiguous().float()
            ctx.grads[id(ctx.logits_perm.raw)] = [ops.logits_permute_grad(dlogits, ctx.perm), True]
        dev = ctx.perm.device
        # Weight gradients are off the critical path (only the optimiser needs them); with RL_SIDE_STREAM=1
        # they run on a side stream beside the dY -> dX chain (everything they read stays referenced until
        # the join below).  Off by default: under hipGraph replay the forks/joins cost more than they hide.
        self._main = torch.cuda.current_stream(dev)
        if self._side is None:
            self._side = torch.cuda.Stream(dev)
        if getattr(ctx, "csr_ready", None) is not None:
            self._main.wait_event(ctx.csr_ready)       # the graph transposes built beside the forward
            ctx.csr_ready = None
        ctx.hold = []
        ctx.bn_pre = {}           # raw tensor -> BatchNorm-backward partials its gradient's producer left (the fused head)
        ctx.bn_done = set()       # raw tensors whose BatchNorm backward already happened (fused at the residual junction)
        # weight-gradient slabs are summed by ONE launch after the last layer (they are only needed by the optimiser)
        # backward finalizes of the virtual rpe stages wait for the next per-point layer's finalize launch and ride along with it
        # (rl_bn_bwd_finalize_batch); the stage's weight-gradient kernel, which needs the result, follows that launch (_flush_rpe)
        ctx.fin_queue, ctx.rpe_after = [], []
        ctx.pending = None if (ops.SIDE_STREAM_WGRAD or ops.NO_DEFERRED_WGRAD) else []
        # ... and the wide layers' weight-gradient KERNELS wait as well: one grouped launch for all of them at the end
        ctx.wbatch = [] if ctx.pending is not None else None
        for rec, lvl in zip(reversed(ctx.tape), reversed(ctx.tape.levels)):
            ops.LEVEL = lvl
            kind = rec[0]
            if kind != "linear":
                self._flush_rpe(ctx)
            if kind == "linear":
                self._bwd_linear(ctx, grads, *rec[1:])
            elif kind == "pool":
                self._bwd_pool(ctx, grads, *rec[1:])
            elif kind == "pool_fused":
                _, name, u, g, csr, idx, pooled, n, d, stage = rec
                if stage:
                    self._bwd_pool_virtual(ctx, grads, name, u, g, csr, idx, pooled, n, d, stage)
                    continue
                GP, init = self._gbuf(ctx, pooled)
                assert init
                gu, gg = self._gbuf(ctx, u), self._gbuf(ctx, g)
                DG = ops.pool_bwd(u, g, idx, self.P[f"{name}.score_fn.0.weight"], n, d, GP, gu[0], gu[1],
                                  grads[f"{name}.score_fn.0.weight"], pending=ctx.pending, batch=ctx.wbatch)
                gu[1] = True
                # gradient of the gather: every gathered point sums its slots in a fixed order
                ops.segment_sum_rows(DG, (0, d // 2), n * self.K, csr, gg[0], g.bstride, accumulate=gg[1])
                gg[1] = True
            elif kind == "add_act":
                _, m2, sc, O = rec
                G, init = self._gbuf(ctx, O)
                assert init
                if not ops.NO_RESID_BN and ops.resid_bn_supported(m2, sc):
                    # the junction's derivative and both BatchNorm backwards behind it in two sweeps
                    g2 = ops.resid_bn_backward(G, O.raw, 0.01, m2, sc, grads[f"{m2.bn}.weight"], grads[f"{m2.bn}.bias"],
                                               grads[f"{sc.bn}.weight"], grads[f"{sc.bn}.bias"], sync=self.sync)
                    ctx.bn_done.update((id(m2.raw), id(sc.raw)))
                else:
                    ops.add_act_bwd(G, O.raw, 0.01)
                    g2 = torch.empty_like(G)
                    ops.copy_rows(G, (0, O.C), O.n, g2, (0, O.C), O.rows, O.n)
                ctx.grads[id(m2.raw)] = [G, True]
                ctx.grads[id(sc.raw)] = [g2, True]
            elif kind == "interp_concat":
                _, prev, skip, csr, catl = rec
                G, init = self._gbuf(ctx, catl)
                assert init
                gp = self._gbuf(ctx, prev)
                ops.segment_sum_rows(G, (0, prev.C), catl.n, csr, gp[0], prev.bstride, accumulate=gp[1])
                gp[1] = True
                if init != "split":              # (else: the skip half went to its gradient directly, _bwd_linear)
                    gs = self._gbuf(ctx, skip)
                    ops.copy_rows(G, (prev.C, skip.C), catl.n, gs[0], (0, skip.C), catl.rows, catl.n, accumulate=gs[1])
                    gs[1] = True
            elif kind == "head":
                _, xh, head, drop = rec
                G, pre = ops.head_bwd(xh, self._w2("fc_end.3.conv.weight"), self.P["fc_end.3.conv.bias"], ctx.perm, head, drop,
                                      grads["fc_end.3.conv.weight"], grads["fc_end.3.conv.bias"], ctx.pending)
                ctx.grads[id(xh.raw)] = [G, True]
                if pre is not None:
                    ctx.bn_pre[id(xh.raw)] = pre
            elif kind == "dropout_philox":
                _, src, dropped, key, p_drop, seed, first_row = rec
                G, init = self._gbuf(ctx, dropped)
                assert init
                ops.dropout_bwd(G, key, seed, p_drop, first_row)
                ctx.grads[id(src.raw)] = [G, True]
            elif kind == "dropout":
                _, src, dropped, mask, scale = rec
                G, init = self._gbuf(ctx, dropped)
                assert init
                ops.scale_mask(G, mask, scale)
                ctx.grads[id(src.raw)] = [G, True]
            else:
                raise AssertionError(kind)
        self._flush_rpe(ctx)
        ops.LEVEL = -1
        self._main.wait_stream(self._side)
        if ctx.pending is not None:
            ops.wgrad_batch_flush(ctx.wbatch)
            ops.wgrad_flush(ctx.pending)
        ctx.tape.clear()
        ctx.grads.clear()
        ctx.keep.clear()
        ctx.hold.clear()

    def _beside(self, ctx: Context, fn, *keep) -> None:
        """Run fn() on the side stream, ordered after everything issued so far on the main stream."""
        if not ops.SIDE_STREAM_WGRAD:
            fn()
            return
        ctx.hold.extend(keep)
        ev = torch.cuda.Event()
        ev.record(self._main)
        self._side.wait_event(ev)
        with torch.cuda.stream(self._side):
            fn()

    def _flush_rpe(self, ctx: Context) -> None:
        """Send out the queued backward finalizes of the virtual rpe stages (if no layer's finalize launch took them along) and
        run what waited for them."""
        if ctx.fin_queue:
            first = ctx.fin_queue.pop(0)
            ops._bn_bwd_finalize(*first, None, also=ctx.fin_queue)
        for fn in ctx.rpe_after:
            fn()
        ctx.rpe_after.clear()

    def _bwd_linear(self, ctx, grads, a, out: Lazy, wname, bname, ks, ns, a_grad):
        G, init = self._gbuf(ctx, out)
        assert init, f"no gradient reached {wname}"
        if out.scale is not None and id(out.raw) not in ctx.bn_done:
            ops.bn_backward(G, out, grads[f"{out.bn}.weight"], grads[f"{out.bn}.bias"], True, sync=self.sync,
                            stats=ctx.bn_pre.pop(id(out.raw), None), also=ctx.fin_queue)
            self._flush_rpe(ctx)
        n_out = out.C
        self._beside(ctx, lambda: ops.wgrad(a, G, out.bstride, n_out, grads[wname], ks, ns,
                                            grads[bname] if bname else None, pending=ctx.pending, batch=ctx.wbatch), G)
        halves = getattr(a, "concat", None) if (a_grad and isinstance(a, Lazy)) else None
        if (halves is not None and not ops.NO_SPLIT_SCATTER and (n_out > 64 or a.C > 64) and id(a.raw) not in ctx.grads
                and halves[1].bstride == halves[1].n and halves[1].raw.shape[0] == a.rows
                and halves[1].raw.shape[1] == halves[1].C and halves[0].C % 4 == 0      # out2 is addressed as a dense (rows, skip.C) tensor
                and not ctx.grads.get(id(halves[1].raw), (None, False))[1]):
            # the decoder's concat [interpolated | skip] (modules.py:362): its gradient leaves the dgrad GEMM in two pieces - the
            # interpolated half to a dense tensor (summed per coarse point by the record behind this one), the skip half
            # straight into the skip tensor's gradient, whose FIRST writer this is (the encoder's own contributions follow) -
            # instead of one (rows, C1 + C2) tensor and a strided copy of its right half
            prev, skip = halves
            gs = self._gbuf(ctx, skip)
            Gp = torch.empty((a.rows, prev.C), dtype=torch.float32, device=G.device)
            gl = Lazy(G, out.B, out.n, out.bstride, n_out)
            ops.gemm(gl, self._w2(wname), ns, ks, a.C, None, out=Gp, out_bstride=a.bstride, out2=gs[0], split_col=prev.C,
                     wsplit=getattr(ctx, "wsplit", None))
            gs[1] = True
            ctx.grads[id(a.raw)] = [Gp, "split"]
            return
        if a_grad and isinstance(a, Lazy):
            ga = self._gbuf(ctx, a)
            if not ga[1]:
                assert a.n == a.bstride and a.raw.shape[0] == a.B * a.n, "first writer must cover the tensor"
            gl = Lazy(G, out.B, out.n, out.bstride, n_out)
            # dA = dY . W^T : the same kernel with the weight strides swapped.  Where this product COMPLETES the gradient of a
            # BatchNorm layer's output - the layer has this one consumer (fc_end.0 -> fc_end.1, the last decoder step -> fc_end.0,
            # pool2.mlp -> mlp2), or this is the second of its two (bn_start -> shortcut, then mlp1 of level 0) - and the streaming
            # kernel takes it, the layer's BatchNorm-backward sums come out of the epilogue (no reduce sweep over G and Y later)
            complete = (self.sync is None and a.scale is not None and a.mean is not None and a.rows == a.raw.shape[0] and
                        ((not ga[1] and (wname in ("fc_end.1.conv.weight", "fc_end.0.conv.weight") or wname.endswith(".mlp2.conv.weight")))
                         or (ga[1] and wname == "encoder.0.mlp1.conv.weight")))
            res = ops.gemm(gl, self._w2(wname), ns, ks, a.C, None, out=ga[0], out_bstride=a.bstride, accumulate=ga[1],
                           wsplit=getattr(ctx, "wsplit", None), bnb=a if complete else None)
            if complete and res[1] is not None:
                ctx.bn_pre[id(a.raw)] = res[1]
            ga[1] = True

    def _bwd_pool_virtual(self, ctx, grads, name, vr, g: Lazy, csr, idx, pooled: Lazy, n, d, stage):
        """Pooling block whose rpe half is virtual, followed by the backward of that stage itself (its BatchNorm + ReLU +
        Linear: mlp_rpe2 for stage 2, mlp_rpe1 for stage 1 - they have no tape records of their own)."""
        h = d // 2
        e = name.rsplit(".", 1)[0]                       # encoder.<l>
        GP, init = self._gbuf(ctx, pooled)
        assert init
        key = ("vgu", id(vr))
        if stage == 2:
            GU = torch.empty((vr.rows, h), dtype=ops.row_dtype(), device=GP.device)     # bf16 in the bf16-storage mode
            first = True
        else:
            GU, first = ctx.grads.pop(key)[0], False     # written by stage 2's Linear backward (dY2 . W2)
        gg = self._gbuf(ctx, g)
        # this launch completes GU (stage 2: its only writer; stage 1: it adds the pooling path to dY2 . W2), so it can
        # leave the batch-statistics sums of the stage's BatchNorm backward as well
        nslots = H.lib().rl_pool_bwd_slots(vr.B * n, d)
        bstats = torch.empty((nslots, 2, h), dtype=torch.float64, device=GP.device)
        DG = ops.pool_bwd(vr, g, idx, self.P[f"{name}.score_fn.0.weight"], n, d, GP, GU, not first,
                          grads[f"{name}.score_fn.0.weight"], pending=ctx.pending, stage=stage, bn_bwd_stats=bstats)
        ops.segment_sum_rows(DG, (0, h), n * self.K, csr, gg[0], g.bstride, accumulate=gg[1])
        gg[1] = True
        # the stage's own backward: batch-statistics terms of its BatchNorm, then weight / bias (/ input) gradients
        layer = f"{e}.mlp_rpe{stage}"
        pending = ctx.pending if ctx.pending is not None else []
        queue = ctx.fin_queue if (self.sync is None and ctx.pending is not None and not ops.NO_BN_BATCH) else None
        coef = ops.rpe_bn_backward(vr, stage, GU, grads[f"{layer}.batch_norm.weight"], grads[f"{layer}.batch_norm.bias"],
                                   sync=self.sync, stats=bstats, nslots=nslots, queue=queue)
        GU1 = torch.empty_like(GU) if stage == 2 else None
        lvl = ops.LEVEL

        def finish():
            keep, ops.LEVEL = ops.LEVEL, lvl
            ops.rpe_wgrad(vr, stage, GU, coef, grads[f"{layer}.conv.weight"], grads[f"{layer}.conv.bias"], pending, GU1)
            ops.LEVEL = keep
            if ctx.pending is None:
                ops.wgrad_flush(pending)
        if queue is not None:
            ctx.rpe_after.append(finish)        # behind the finalize launch that carries this stage's sums (_flush_rpe)
        else:
            finish()
        if stage == 2:
            ctx.grads[key] = [GU1, True]

    def _bwd_pool(self, ctx, grads, name, u: Lazy, g: Lazy, csr, X, S, pooled: Lazy, n, d):
        B, K, h = u.B, self.K, d // 2
        rows = B * n * K
        GP, init = self._gbuf(ctx, pooled)
        assert init
        dS, dX = ops.attpool_bwd(X, S, pooled.raw, GP, B * n, K)
        Ws = self.P[f"{name}.score_fn.0.weight"]
        self._beside(ctx, lambda: ops.wgrad(ops.plain(X, B, n * K), dS, n * K, d,
                                            grads[f"{name}.score_fn.0.weight"], 1, d, None, pending=ctx.pending, batch=ctx.wbatch), X, dS)
        gu = self._gbuf(ctx, u)
        gg = self._gbuf(ctx, g)
        if ops.NO_SPLIT_SCATTER or d <= 64:      # the epilogue lives in the wide (LDS-tiled) kernel only
            ops.gemm(ops.plain(dS, B, n * K), Ws, d, 1, d, None, out=dX, out_bstride=n * K, accumulate=True,
                     wsplit=getattr(ctx, "wsplit", None))
            ops.copy_rows(dX, (0, h), n * K, gu[0], (0, h), rows, n * K, accumulate=gu[1])
            ops.segment_sum_rows(dX, (h, h), n * K, csr, gg[0], g.bstride, accumulate=gg[1])
        else:
            # dX = dP*A + dS.W leaves the GEMM epilogue in two pieces: the rpe-branch half is stored (or accumulated)
            # where it belongs, the gathered half goes to a dense tensor that is then summed per gathered point
            DG = torch.empty((rows, h), dtype=torch.float32, device=dX.device)
            ops.gemm(ops.plain(dS, B, n * K), Ws, d, 1, d, None, out=gu[0], out_bstride=n * K, accumulate=gu[1],
                     addend=dX, out2=DG, split_col=h, wsplit=getattr(ctx, "wsplit", None))
            ops.segment_sum_rows(DG, (0, h), n * K, csr, gg[0], g.bstride, accumulate=gg[1])
        gu[1] = gg[1] = True
