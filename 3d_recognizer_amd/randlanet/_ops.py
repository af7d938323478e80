"""Tensor-level wrappers over the C ABI (include/rl_randlanet.h).  PyTorch is used for device
memory and streams only; all arithmetic happens in librandla_hip.so.  Every wrapper checks
shapes on the host before launching (a faulting kernel can reset the GPU) and raises
HipKernelError on failure - there is no fallback implementation.
"""
import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _hip as H

F32 = torch.float32
BF16 = torch.bfloat16

# Storage of the neighbourhood-row tensors that only exist between two kernels of the backward pass (rl_randlanet.h
# "bf16-storage mode"): "f32" (default, the parity mode) or "bf16" - GU / DG of the fused pooling blocks and, on the
# 128-wide level, X / dS are then written and read as bf16 (half the bytes of the largest tensors of a step); coordinates,
# neighbour search, BatchNorm statistics, softmax, every accumulator, parameters and Adam stay fp32.
_STORAGE = __import__("os").environ.get("RL_STORAGE", "f32")


def set_storage(mode: str) -> None:
    global _STORAGE
    if mode not in ("f32", "bf16"):
        raise ValueError(f"storage must be 'f32' or 'bf16', got {mode!r}")
    if mode == "bf16" and get_wide_gemm() == "fp32":
        raise H.HipKernelError("bf16 storage needs the bf16x3 / bf16 arithmetic mode (RL_WIDE_GEMM)")
    _STORAGE = mode


def get_storage() -> str:
    return _STORAGE


def row_dtype() -> torch.dtype:
    """dtype of the (points*K)-row gradient tensors between backward kernels."""
    return BF16 if _STORAGE == "bf16" else F32


@dataclass
class Lazy:
    """A (rows, C) fp32 tensor whose value is act(raw*scale + shift) - the output of a
    SharedMLP before its BatchNorm+activation is applied (reference modules.py:93-104)."""
    raw: torch.Tensor                      # storage; logical rows are (B, n) with batch stride
    B: int
    n: int
    bstride: int                           # rows between clouds in `raw`
    C: int
    scale: Optional[torch.Tensor] = None
    shift: Optional[torch.Tensor] = None
    act: int = H.ACT_NONE
    slope: float = 0.0
    mean: Optional[torch.Tensor] = None    # saved batch statistics (training)
    invstd: Optional[torch.Tensor] = None
    bn: Optional[str] = None               # parameter prefix of its BatchNorm, if any

    @property
    def rows(self) -> int:
        return self.B * self.n

    def prefix(self, n: int) -> "Lazy":
        """First n rows of every cloud (the reference's random-sampling prefix slice)."""
        assert n <= self.n
        return Lazy(self.raw, self.B, n, self.bstride, self.C, self.scale, self.shift, self.act,
                    self.slope, self.mean, self.invstd, self.bn)


def plain(t: torch.Tensor, B: int, n: int) -> Lazy:
    assert t.dim() == 2 and t.shape[0] == B * n and t.is_contiguous()
    return Lazy(t, B, n, n, t.shape[1])


def _dev_check(*ts):
    """Every operand of a launch must be a contiguous tensor on the CURRENT HIP device: kernels go to the current
    device's current stream (H.stream_ptr), so a tensor living on another GPU would be dereferenced by the wrong
    device - a memory fault on a multi-GPU node.  Raised before anything is launched."""
    cur = H.current_device()
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise H.HipKernelError("HIP kernels need device tensors")
        if t.get_device() != cur:
            raise H.HipKernelError(
                f"tensor on cuda:{t.get_device()} but the current device is cuda:{cur}: run under "
                "`with torch.cuda.device(tensor.device)` (launches use the current device's stream)")
        if not t.is_contiguous():
            raise H.HipKernelError("HIP kernels need contiguous tensors")


def _st():
    return H.stream_ptr()


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg).
    Each record: (category, shape key, algorithmic bytes, flops, start event, end event)."""

    def __init__(self):
        self.records = []

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for cat, key, nbytes, flops, e0, e1, _kern, _lvl in self.records:
            a = agg.setdefault((cat, key), dict(category=cat, shape=key, launches=0, ms=0.0, bytes=0, flops=0))
            a["launches"] += 1
            a["ms"] += e0.elapsed_time(e1)
            a["bytes"] += nbytes
            a["flops"] += flops
        return sorted(agg.values(), key=lambda a: -a["ms"])


TIMER: Optional[KernelTimer] = None
LEVEL = -1      # encoder level the engine is working on (-1: outside the encoder) - a tag on the timer's records only
# Weight gradients on a second stream beside the dY->dX chain: measured 11.2 -> 19-21 ms per step under
# hipGraph replay (every fork/join becomes a cross-branch dependency in the graph), so OFF by default.
NO_BN_SMALL = bool(int(__import__("os").environ.get("RL_NO_BN_SMALL", "0")))      # A/B: small tensors take the three-launch path too
SIDE_STREAM_WGRAD = bool(int(__import__("os").environ.get("RL_SIDE_STREAM", "0")))
NO_FUSED_POOL = bool(int(__import__("os").environ.get("RL_NO_FUSED_POOL", "0")))         # diagnostics only
NO_DEFERRED_WGRAD = bool(int(__import__("os").environ.get("RL_NO_DEFERRED_WGRAD", "0")))  # diagnostics only
NO_SPLIT_SCATTER = bool(int(__import__("os").environ.get("RL_NO_SPLIT_SCATTER", "0")))   # diagnostics only
NO_RESID_BN = bool(int(__import__("os").environ.get("RL_NO_RESID_BN", "0")))             # diagnostics only
NO_RPE_TENSOR = bool(int(__import__("os").environ.get("RL_NO_RPE_TENSOR", "0")))         # diagnostics only
# the rpe branch (mlp_rpe1 / mlp_rpe2 outputs) recomputed inside its consumers instead of stored, where the fused pooling
# kernels support it (16 neighbours, d <= 64); RL_NO_VIRTUAL_RPE=1 keeps the tensors (diagnostics / cross-checks)
VIRTUAL_RPE = not bool(int(__import__("os").environ.get("RL_NO_VIRTUAL_RPE", "0")))
FORCE_BRUTE_KNN = bool(int(__import__("os").environ.get("RL_KNN_BRUTE", "0")))         # diagnostics only
DEBUG_SYNC = bool(int(__import__("os").environ.get("RL_DEBUG_SYNC", "0")))   # print + sync around every launch


class _rec:
    """with _rec(category, key, bytes, flops): <one launch>  - free unless TIMER is set."""
    __slots__ = ("args", "e0")

    def __init__(self, cat, key, nbytes, flops=0):
        self.args = (cat, key, nbytes, flops)

    def __enter__(self):
        if DEBUG_SYNC:
            print("launch", self.args[0], self.args[1], flush=True)
        if TIMER is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if DEBUG_SYNC:
            torch.cuda.synchronize()
        if TIMER is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            kern = H.lib().rl_last_kernel().decode()
            TIMER.records.append(self.args + (self.e0, e1, kern, LEVEL))
        return False


# ------------------------------------------------------------------------------------- knn
def _knn_workspace(device, B: int, Ns: int, Nq: int, k: int, brute: bool):
    """Scratch for the grid search (None -> tiled brute force inside the library)."""
    if brute or FORCE_BRUTE_KNN or k > H.KNN_MAX_K or Ns < k:
        return None, 0
    nbytes = H.lib().rl_knn_workspace_bytes(B, Ns, Nq, k)
    if nbytes <= 0:
        return None, 0
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


def knn_i32(support: torch.Tensor, query: torch.Tensor, Ns: int, Nq: int, k: int, brute: bool = False
            ) -> Tuple[torch.Tensor, torch.Tensor]:
    """support (B, >=Ns, 3), query (B, >=Nq, 3): searches the first Ns / Nq points of each cloud."""
    _dev_check(support, query)
    assert support.dtype == F32 and query.dtype == F32 and support.shape[-1] == 3 and query.shape[-1] == 3
    B = support.shape[0]
    assert query.shape[0] == B and support.shape[1] >= Ns and query.shape[1] >= Nq
    idx = torch.empty((B, Nq, k), dtype=torch.int32, device=support.device)
    d2 = torch.empty((B, Nq, k), dtype=F32, device=support.device)
    ws, nbytes = _knn_workspace(support.device, B, Ns, Nq, k, brute)
    with _rec("knn", (B, Ns, Nq, k), B * (12 * (Ns + Nq) + 8 * Nq * k), 8 * B * Ns * Nq):
        H.check(H.lib().rl_knn_i32(support.data_ptr(), support.shape[1], query.data_ptr(), query.shape[1],
                                   B, Ns, Nq, k, idx.data_ptr(), d2.data_ptr(), H.ptr(ws), nbytes, _st()),
                "rl_knn_i32")
    return idx, d2


def knn_multi(xyz: torch.Tensor, tasks):
    """Several prefix searches over the same (B, N, 3) clouds in one launch set: tasks is a list of
    (Ns, Nq, k) - support = first Ns points, queries = first Nq points of every cloud.
    Returns [(idx int32 (B,Nq,k), d2 fp32 (B,Nq,k)), ...]."""
    _dev_check(xyz)
    assert xyz.dtype == F32 and xyz.dim() == 3 and xyz.shape[2] == 3
    B, N, _ = xyz.shape
    out = []
    for c0 in range(0, len(tasks), H.KNN_MAX_TASKS):
        chunk = tasks[c0:c0 + H.KNN_MAX_TASKS]
        arr = (H.KnnTask * len(chunk))()
        res = []
        nbytes_io = 0
        for t, (Ns, Nq, k) in zip(arr, chunk):
            assert 0 < k <= Ns <= N and 0 < Nq <= N and k <= H.KNN_MAX_K
            idx = torch.empty((B, Nq, k), dtype=torch.int32, device=xyz.device)
            d2 = torch.empty((B, Nq, k), dtype=F32, device=xyz.device)
            t.support, t.support_bstride, t.query, t.query_bstride = xyz.data_ptr(), N, xyz.data_ptr(), N
            t.Ns, t.Nq, t.k, t.idx_out, t.d2_out = Ns, Nq, k, idx.data_ptr(), d2.data_ptr()
            res.append((idx, d2))
            nbytes_io += B * (12 * (Ns + Nq) + 8 * Nq * k)
        nbytes = H.lib().rl_knn_multi_workspace_bytes(arr, len(chunk), B)
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=xyz.device)
        with _rec("knn_multi", tuple(chunk), nbytes_io, sum(8 * B * a * b for a, b, _ in chunk)):
            H.check(H.lib().rl_knn_multi(arr, len(chunk), B, ws.data_ptr(), ws.numel(), _st()), "rl_knn_multi")
        out.extend(res)
    return out


def knn_f32(support: torch.Tensor, query: torch.Tensor, k: int, brute: bool = False
            ) -> Tuple[torch.Tensor, torch.Tensor]:
    _dev_check(support, query)
    B, Ns, _ = support.shape
    Nq = query.shape[1]
    idx = torch.empty((B, Nq, k), dtype=torch.int64, device=support.device)
    d2 = torch.empty((B, Nq, k), dtype=F32, device=support.device)
    ws, nbytes = _knn_workspace(support.device, B, Ns, Nq, k, brute)
    with _rec("knn", (B, Ns, Nq, k), B * (12 * (Ns + Nq) + 12 * Nq * k), 8 * B * Ns * Nq):
        H.check(H.lib().rl_knn_f32(support.data_ptr(), query.data_ptr(), B, Ns, Nq, k, idx.data_ptr(),
                                   d2.data_ptr(), H.ptr(ws), nbytes, _st()), "rl_knn_f32")
    return idx, d2


# ------------------------------------------------------------------------------------ gemm
@dataclass
class Rpe:
    """Relative-position-encoding A operand (reference modules.py:173-186), never materialised."""
    xyz: torch.Tensor      # (B, n_parent, 3)
    idx: torch.Tensor      # (B, n, K) int32
    d2: torch.Tensor       # (B, n, K) fp32
    B: int
    n: int
    K: int

    @property
    def rows(self) -> int:
        return self.B * self.n * self.K


def set_wide_gemm(mode: str) -> None:
    """Arithmetic of the wide GEMM / weight-gradient kernels: "bf16x3" (default), "fp32" or "bf16" (rl_set_wide_gemm)."""
    H.check(H.lib().rl_set_wide_gemm(mode.encode()), "rl_set_wide_gemm")


def get_wide_gemm() -> str:
    return H.lib().rl_get_wide_gemm().decode()


def set_wgemm_staging(how: str) -> None:
    """Operand staging of the wide GEMM in the bf16 modes: "registers" or "dma" (rl_set_wgemm_staging); same results."""
    H.check(H.lib().rl_set_wgemm_staging(how.encode()), "rl_set_wgemm_staging")


def set_wgemm_tile(how: str) -> None:
    """Output tile of the LDS-DMA wide GEMM: "auto" (64-row / 64-column tiles for launches with few rows) or "128"; same Y."""
    H.check(H.lib().rl_set_wgemm_tile(how.encode()), "rl_set_wgemm_tile")


def set_gemm_ksplit(enable: bool) -> None:
    """Diagnostics: the K split of wide products with few output tiles on / off (rl_set_gemm_ksplit)."""
    H.check(H.lib().rl_set_gemm_ksplit(int(bool(enable))), "rl_set_gemm_ksplit")


def rpe_build(a: "Rpe", distances: bool = False) -> Lazy:
    """The relative position encoding of every neighbourhood row, written out once (rows x 12 floats: 10 channels +
    2 of padding) so that mlp_rpe1's forward and weight gradient read a plain tensor (modules.py:173-186)."""
    _dev_check(a.xyz, a.idx, a.d2)
    assert a.idx.dtype == torch.int32 and a.idx.shape == (a.B, a.n, a.K) and a.d2.shape == a.idx.shape
    assert a.xyz.shape[0] == a.B and a.xyz.shape[1] >= a.n and a.xyz.shape[2] == 3
    out = torch.empty((a.rows, 12), dtype=F32, device=a.xyz.device)
    with _rec("rpe_build", (a.rows,), (48 + 8) * a.rows, 0):
        fn = H.lib().rl_rpe_build_dist if distances else H.lib().rl_rpe_build     # a.d2 holds distances, not squares
        H.check(fn(a.xyz.data_ptr(), a.xyz.shape[1], a.idx.data_ptr(), a.d2.data_ptr(), a.B, a.n, a.K,
                   out.data_ptr(), _st()), "rl_rpe_build")
    return Lazy(out, a.B, a.n * a.K, a.n * a.K, 10)


def _fill_a(d, a):
    if isinstance(a, Rpe):
        _dev_check(a.xyz, a.idx, a.d2)
        assert a.idx.dtype == torch.int32 and a.idx.shape == (a.B, a.n, a.K) and a.d2.shape == a.idx.shape
        assert a.xyz.shape[0] == a.B and a.xyz.shape[1] >= a.n and a.xyz.shape[2] == 3
        d.a_mode = 1
        d.xyz, d.xyz_bstride = a.xyz.data_ptr(), a.xyz.shape[1]
        d.nbr_idx, d.nbr_d2, d.nbr_k = a.idx.data_ptr(), a.d2.data_ptr(), a.K
        d.B, d.n, d.K = a.B, a.n, 10
        return a.rows, 10
    _dev_check(a.raw, a.scale, a.shift)
    assert a.raw.dtype in (F32, BF16) and a.raw.dim() == 2 and a.raw.shape[1] >= a.C
    assert a.raw.shape[0] >= (a.B - 1) * a.bstride + a.n, "A operand rows out of range"
    d.a_mode = 0
    d.A, d.lda, d.a_bstride = a.raw.data_ptr(), a.raw.shape[1], a.bstride
    if a.scale is not None:
        assert a.scale.numel() == a.C and a.shift.numel() == a.C
        d.in_scale, d.in_shift, d.in_act, d.in_slope = a.scale.data_ptr(), a.shift.data_ptr(), a.act, a.slope
    d.B, d.n, d.K = a.B, a.n, a.C
    return a.rows, a.C


def weight_strides(W: torch.Tensor, transposed: bool, K: int, N: int) -> Tuple[int, int]:
    """(w_ks, w_ns) for a reference-layout weight: Conv2d/Linear (N,K[,1,1]) or
    ConvTranspose2d (K,N,1,1)."""
    if transposed:
        assert W.shape[0] == K and W.shape[1] == N, (tuple(W.shape), K, N)
        return N, 1
    assert W.shape[0] == N and W.shape[1] == K, (tuple(W.shape), K, N)
    return 1, K


# weights of the wide layers pre-split into bf16 head / tail planes, per orientation: split_weights() returns a dict
# (data_ptr, w_ks, w_ns, K, N) -> planes that the engine hands to gemm(wsplit=...) for the products of that pass only
NO_WSPLIT = bool(int(__import__("os").environ.get("RL_NO_WSPLIT", "0")))     # diagnostics: keep the 4-wavefront wide GEMM


def split_weights(entries) -> dict:
    """entries: list of (W, w_ks, w_ns, K, N[, True]).  One launch; returns {(data_ptr, w_ks, w_ns, K, N): planes}.  Products with
    N <= 64 and K <= 64 run on the streaming kernels and get no planes - unless the entry carries a sixth element (the narrow half of a
    gemm_pair)."""
    out_map = {}
    if NO_WSPLIT or get_wide_gemm() == "fp32" or not entries:
        return out_map
    entries = [e[:5] for e in entries if e[3] % 8 == 0 and (e[4] > 64 or e[3] > 64 or len(e) > 5)]
    if not entries:
        return out_map
    total = sum(2 * K * N + 8 for _, _, _, K, N in entries)
    buf = torch.empty(total, dtype=torch.bfloat16, device=entries[0][0].device)
    arr = (H.WsplitItem * len(entries))()
    off = 0
    for it, (W, ks, ns, K, N) in zip(arr, entries):
        _dev_check(W)
        assert W.dtype == F32 and W.numel() == K * N
        out = buf[off:off + 2 * K * N]
        it.W, it.w_ks, it.w_ns, it.K, it.N, it.out = W.data_ptr(), ks, ns, K, N, out.data_ptr()
        out_map[(W.data_ptr(), ks, ns, K, N)] = out
        off += (2 * K * N + 7) // 8 * 8           # 16-byte aligned planes
    with _rec("split_weights", (len(entries),), 6 * sum(K * N for _, _, _, K, N in entries), 0):
        H.check(H.lib().rl_split_weights(arr, len(entries), _st()), "rl_split_weights")
    return out_map


def gemm_stat_slots(M: int, N: int, K: int) -> int:
    """Slots of `stats` that gemm(..., stats=...) fills for an (M, K) x (K, N) product - the count bn_finalize must be given."""
    return int(H.lib().rl_gemm_stat_slots(M, N, K))


def gemm(a, W: torch.Tensor, w_ks: int, w_ns: int, N: int, bias: Optional[torch.Tensor] = None, *,
         out: Optional[torch.Tensor] = None, out_bstride: Optional[int] = None,
         accumulate: bool = False, stats: Optional[torch.Tensor] = None,
         addend: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None,
         out2_index: Optional[torch.Tensor] = None, out2_bstride: int = 0, split_col: int = 0,
         wsplit: Optional[dict] = None, pivot: Optional[tuple] = None, bnb: Optional[Lazy] = None):
    """Y = A'.W (+ bias).  With `out2` (split epilogue, wide layers only): v = A'.W + addend; columns < split_col go to
    `out` (which then has split_col columns), the others to the dense (M, N - split_col) tensor `out2` (or, with
    `out2_index`, atomically to the rows it names) - the two halves of a concat's gradient in one pass.
    pivot (with stats): (running_mean, conv bias left out of this product or None) - the statistics are SHIFTED sums around
    running_mean - bias (rl_gemm_desc.stats_pivot_*); the matching bn_finalize call must say pivoted=True.
    bnb (a Lazy with batch statistics): this product is the complete gradient w.r.t. `bnb`'s ACTIVATED value and the streaming
    kernel takes it - the BatchNorm-backward sums of `bnb`'s layer come out as a by-product (rl_gemm_desc.bnb_*): returns
    (out, (partials, nslots)) with the pair bn_backward(stats=...) takes; (out, None) when the product does not qualify."""
    d = H.GemmDesc()
    M, K = _fill_a(d, a)
    assert isinstance(a, Rpe) or a.raw.dtype == F32, "rl_gemm reads fp32 rows"
    rows_per_batch = a.n * a.K if isinstance(a, Rpe) else a.n
    _dev_check(W, bias, out, stats)
    assert W.dtype == F32 and W.numel() == K * N
    if out is None:
        out = torch.empty((M, N), dtype=F32, device=W.device)
        out_bstride = rows_per_batch
    else:
        assert out.dtype == F32 and out.dim() == 2 and out.shape[1] >= (split_col if out2 is not None else N)
        out_bstride = rows_per_batch if out_bstride is None else out_bstride
        assert out.shape[0] >= (d.B - 1) * out_bstride + rows_per_batch, "Y rows out of range"
    if bias is not None:
        assert bias.numel() == N
    if stats is not None:
        assert stats.dtype == torch.float64 and stats.numel() >= gemm_stat_slots(M, N, K) * 2 * N
    d.N, d.W, d.w_ks, d.w_ns, d.bias = N, W.data_ptr(), w_ks, w_ns, H.ptr(bias)
    planes = wsplit.get((W.data_ptr(), w_ks, w_ns, K, N)) if wsplit else None
    if planes is not None:
        d.W_split = planes.data_ptr()
    d.Y, d.ldy, d.y_bstride, d.accumulate = out.data_ptr(), out.shape[1], out_bstride, int(accumulate)
    d.stats = H.ptr(stats)
    if pivot is not None:
        assert stats is not None and pivot[0].numel() == N and (pivot[1] is None or pivot[1].numel() == N)
        _dev_check(*pivot)
        d.stats_pivot_mean, d.stats_pivot_bias = pivot[0].data_ptr(), H.ptr(pivot[1])
    if addend is not None or out2 is not None:
        _dev_check(addend, out2, out2_index)
        assert addend is None or (addend.dtype == F32 and addend.shape == (M, N))
        if out2 is not None:
            assert out2.dtype == F32 and 0 < split_col < N and out2.shape[1] == N - split_col
            if out2_index is None:
                assert out2.shape[0] >= M
            else:
                assert out2_index.dtype == torch.int32 and out2_index.numel() == M
                assert out2.shape[0] >= (d.B - 1) * out2_bstride + 1
        d.addend, d.out2, d.out2_index = H.ptr(addend), H.ptr(out2), H.ptr(out2_index)
        d.out2_bstride, d.split_col = out2_bstride, split_col
    bnb_pre = None
    if bnb is not None:
        if (stats is None and pivot is None and bnb.mean is not None and bnb.scale is not None and bnb.raw.dtype == F32
                and bnb.raw.shape[1] == out.shape[1] and bnb.bstride == out_bstride and bnb.C == N and not NO_BNB_EPILOGUE
                and 4 * M * N <= BNB_MAX_BYTES and H.lib().rl_gemm_streams(C.byref(d))):
            _dev_check(bnb.raw, bnb.scale, bnb.shift, bnb.mean, bnb.invstd)
            st_b = new_stats(W.device, N)
            d.stats = st_b.data_ptr()
            d.bnb_Y, d.bnb_scale, d.bnb_shift = bnb.raw.data_ptr(), bnb.scale.data_ptr(), bnb.shift.data_ptr()
            d.bnb_mean, d.bnb_invstd, d.bnb_act, d.bnb_slope = bnb.mean.data_ptr(), bnb.invstd.data_ptr(), bnb.act, bnb.slope
            bnb_pre = (st_b, gemm_stat_slots(M, N, K))
    kfloats = H.lib().rl_gemm_kslab_floats(M, N, K) if (N > 64 and not isinstance(a, Rpe) and out2 is None and addend is None) else 0
    if kfloats > 0:
        kslab = _slab(W.device, kfloats)
        d.kslab, d.kslab_floats = kslab.data_ptr(), kslab.numel()
    with _rec("gemm_rpe" if isinstance(a, Rpe) else "gemm", (M, K, N), 4 * (M * (K if not isinstance(a, Rpe) else 6) + M * N * (2 if accumulate else 1) + K * N), 2 * M * K * N):
        H.check(H.lib().rl_gemm(C.byref(d), _st()), "rl_gemm")
    return (out, bnb_pre) if bnb is not None else out


# A/B: the BatchNorm-backward sums as a by-product of the streaming input-gradient GEMM (round 6); off = a reduce sweep per layer
NO_BNB_EPILOGUE = bool(int(__import__("os").environ.get("RL_NO_BNB_EPILOGUE", "0")))
# ... only for tensors up to this size: the epilogue reads the layer's output 4 bytes per lane, a reduce sweep 16 - on a large
# tensor that costs more than the launch it saves (fc_end.0 at 8 clouds: 84 MB, +13 us per step; measured)
BNB_MAX_BYTES = int(__import__("os").environ.get("RL_BNB_MAX_BYTES", str(32 << 20)))
NO_GEMM_PAIR = bool(int(__import__("os").environ.get("RL_NO_GEMM_PAIR", "0")))      # A/B: mlp1 / shortcut as two launches


def gemm_pair(a, first: tuple, second: tuple, wsplit: Optional[dict]):
    """Y1 = A'.W1 and Y2 = A'.W2 in ONE launch (rl_gemm_pair: mlp1 + shortcut of an encoder level, which share their input).
    first / second: (W, w_ks, w_ns, N, stats or None, pivot tuple or None).  Returns (Y1, Y2), or None when the pair cannot go out
    as one launch (the caller then issues two gemm() calls).  Statistics: H.row_blocks(M, 128) slots each."""
    if NO_GEMM_PAIR or not wsplit or isinstance(a, Rpe) or a.raw.dtype != F32:
        return None
    descs, outs = [], []
    M = K = None
    for (W, w_ks, w_ns, N, stats, pivot) in (first, second):
        d = H.GemmDesc()
        M, K = _fill_a(d, a)
        planes = wsplit.get((W.data_ptr(), w_ks, w_ns, K, N))
        if planes is None:
            return None
        _dev_check(W, stats)
        out = torch.empty((M, N), dtype=F32, device=W.device)
        d.N, d.W, d.w_ks, d.w_ns, d.W_split = N, W.data_ptr(), w_ks, w_ns, planes.data_ptr()
        d.Y, d.ldy, d.y_bstride = out.data_ptr(), N, a.n
        if stats is not None:
            assert stats.dtype == torch.float64 and stats.numel() >= H.row_blocks(M, 128) * 2 * N
            d.stats = stats.data_ptr()
            if pivot is not None:
                _dev_check(*pivot)
                d.stats_pivot_mean, d.stats_pivot_bias = pivot[0].data_ptr(), H.ptr(pivot[1])
        descs.append(d)
        outs.append(out)
    if not H.lib().rl_gemm_pair_supported(C.byref(descs[0]), C.byref(descs[1])):
        return None
    N1, N2 = first[3], second[3]
    with _rec("gemm", (M, K, N1 + N2), 4 * (M * K + M * (N1 + N2) + K * (N1 + N2)), 2 * M * K * (N1 + N2)):
        H.check(H.lib().rl_gemm_pair(C.byref(descs[0]), C.byref(descs[1]), _st()), "rl_gemm_pair")
    return outs[0], outs[1]


_SLAB = {}


def _slab(device, floats: int) -> torch.Tensor:
    """Scratch for partial weight-gradient slabs: one growing buffer per (device, stream), reused in
    stream order (launches on one stream serialise, so consecutive users cannot overlap)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _SLAB.get(key)
    if buf is None or buf.numel() < floats:
        buf = torch.empty(max(floats, 1 << 20), dtype=F32, device=device)
        _SLAB[key] = buf
    return buf


NO_WGRAD_BATCH = bool(int(__import__("os").environ.get("RL_NO_WGRAD_BATCH", "0")))      # diagnostics: one launch per layer


WGRAD_BATCH_BYTES = int(__import__("os").environ.get("RL_WGRAD_BATCH_BYTES", str(3 << 30)))   # operands a queued group may pin


def wgrad(a, dY: torch.Tensor, dy_bstride: int, N: int, dW: torch.Tensor, w_ks: int, w_ns: int,
          dbias: Optional[torch.Tensor] = None, pending: Optional[list] = None, batch: Optional[list] = None) -> None:
    """dW / dbias of a layer.  With `pending` (a list) only the per-workgroup partial slabs are produced, in a
    slab of their own, and the layer is queued for `wgrad_flush` - one reduction launch for a whole backward.
    With `batch` (a list, needs `pending`) a wide layer is not even launched here: its descriptor is queued for
    `wgrad_batch_flush`, which runs all of them as ONE grouped launch (they are independent of each other and small)."""
    d = H.WgradDesc()
    M, K = _fill_a(d, a)
    rows_per_batch = a.n * a.K if isinstance(a, Rpe) else a.n
    _dev_check(dY, dW, dbias)
    rows_bf16 = dY.dtype == BF16
    assert dY.dtype in (F32, BF16) and dY.dim() == 2 and dY.shape[1] >= N
    assert isinstance(a, Rpe) or a.raw.dtype == dY.dtype, "A and dY must share their storage type"
    d.rows_bf16 = int(rows_bf16)
    assert dY.shape[0] >= (d.B - 1) * dy_bstride + rows_per_batch
    assert dW.numel() == K * N and (dbias is None or dbias.numel() == N)
    floats = H.lib().rl_wgrad_slab_floats(M, N, K)
    slab = _slab(dY.device, floats) if pending is None else torch.empty(floats, dtype=F32, device=dY.device)
    d.N, d.dY, d.lddy, d.dy_bstride = N, dY.data_ptr(), dY.shape[1], dy_bstride
    d.dW, d.w_ks, d.w_ns, d.dbias = dW.data_ptr(), w_ks, w_ns, H.ptr(dbias)
    d.slab, d.slab_floats = slab.data_ptr(), slab.numel()
    d.defer_reduce = 0 if pending is None else 1
    es = 2 if rows_bf16 else 4
    nbytes, flops = es * (M * (K if not isinstance(a, Rpe) else 6) + M * N) + 4 * K * N, 2 * M * K * N
    kind = H.lib().rl_wgrad_batchable(C.byref(d)) if (batch is not None and pending is not None and not NO_WGRAD_BATCH) else 0
    if kind:
        # (the operands stay referenced by the queue entry until the grouped launch has been issued: the queue is flushed
        # early once it holds WGRAD_BATCH_BYTES of them, so that a large configuration does not keep every layer's A / dY
        # alive until the end of backward - a handful of grouped launches still remove most of the launch overhead)
        batch.append((d, nbytes, flops, a, dY, slab, kind))
        if sum(b[1] for b in batch) > WGRAD_BATCH_BYTES:
            wgrad_batch_flush(batch)
    else:
        with _rec("wgrad_rpe" if isinstance(a, Rpe) else "wgrad", (M, K, N), nbytes, flops):
            H.check(H.lib().rl_wgrad(C.byref(d), _st()), "rl_wgrad")
    if pending is not None:
        it = H.WgradReduceItem()
        it.slab, it.dW, it.dbias, it.w_ks, it.w_ns = slab.data_ptr(), dW.data_ptr(), H.ptr(dbias), w_ks, w_ns
        it.nsplit, it.N, it.K = H.lib().rl_wgrad_nsplit(M, N, K), N, K
        pending.append((it, slab, dW, dbias))


def wgrad_batch_flush(batch: list) -> None:
    """The queued weight gradients as grouped launches (rl_wgrad_batch): one for the wide layers, one for the narrow
    (streaming) ones; before `wgrad_flush`."""
    for kind, name in ((1, "wgrad_batch"), (2, "swgrad_batch")):
        part = [b for b in batch if b[6] == kind]
        if not part:
            continue
        arr = (H.WgradDesc * len(part))(*[b[0] for b in part])
        with _rec(name, (len(part),), sum(b[1] for b in part), sum(b[2] for b in part)):
            H.check(H.lib().rl_wgrad_batch(arr, len(part), _st()), "rl_wgrad_batch")
    batch.clear()


def wgrad_flush(pending: list) -> None:
    """Sum the partial slabs of every queued layer (fixed order, deterministic) in one launch per 48 layers."""
    if not pending:
        return
    arr = (H.WgradReduceItem * len(pending))(*[p[0] for p in pending])
    nbytes = sum(4 * p[1].numel() for p in pending)
    with _rec("wgrad_reduce_batch", (len(pending),), nbytes, 0):
        H.check(H.lib().rl_wgrad_reduce_batch(arr, len(pending), _st()), "rl_wgrad_reduce_batch")
    pending.clear()


# -------------------------------------------------------------------------------------- bn
def new_stats(device, C: int) -> torch.Tensor:
    return torch.empty((H.MAX_SLOTS, 2, C), dtype=torch.float64, device=device)


class SyncGroup:
    """Data-parallel EQUIVALENCE mode: BatchNorm batch statistics and the loss' class sums are
    all-reduced over the ranks, so that N ranks on shards of a batch compute exactly the single-process step on the whole
    batch (up to fp32 summation order).  Not the throughput path: ~150 small collectives per step, no hipGraph.
    `staged=True` moves the (tiny, float64) records through host memory - for backends without device collectives
    (gloo rehearsals on a one-GPU box)."""

    def __init__(self, world: int, group=None, staged: bool = False):
        self.world, self.group, self.staged = world, group, staged
        # where this rank's shard sits in the global batch (set_shard); until then: `world` equal shards, rank unknown
        self.local_clouds, self.global_clouds, self.cloud_offset = 1, world, 0

    def set_shard(self, local_clouds: int, rank: Optional[int] = None) -> None:
        """Tell the group how many clouds this rank holds; the ranks exchange their counts (one tiny host all-reduce), so
        that shards of DIFFERENT sizes (a batch the world size does not divide) still normalise by the global row count
        and know their offset in the batch (Dropout mask slices)."""
        import torch.distributed as dist
        if self.world <= 1 or not (dist.is_available() and dist.is_initialized()):
            self.local_clouds, self.global_clouds, self.cloud_offset = local_clouds, local_clouds * self.world, 0
            return
        rank = dist.get_rank(self.group) if rank is None else rank
        counts = torch.zeros(self.world, dtype=torch.int64)
        counts[rank] = local_clouds
        if dist.get_backend(self.group) == "nccl":
            dev_counts = counts.cuda()
            dist.all_reduce(dev_counts, op=dist.ReduceOp.SUM, group=self.group)
            counts = dev_counts.cpu()
        else:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)
        self.local_clouds, self.global_clouds = local_clouds, int(counts.sum())
        self.cloud_offset = int(counts[:rank].sum())

    def global_rows(self, rows: int) -> int:
        """Rows of the GLOBAL batch for a tensor that has `rows` rows on this rank (every such tensor has the same number
        of rows per cloud on every rank)."""
        assert rows % self.local_clouds == 0, (rows, self.local_clouds)
        return rows // self.local_clouds * self.global_clouds

    def allreduce(self, t: torch.Tensor) -> None:
        import torch.distributed as dist
        if self.staged:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)


def _stats_totals(stats: torch.Tensor, nslots: int, C: int) -> torch.Tensor:
    out = torch.empty((1, 2, C), dtype=torch.float64, device=stats.device)
    H.check(H.lib().rl_bn_reduce_slots(stats.data_ptr(), nslots, C, out.data_ptr(), _st()), "rl_bn_reduce_slots")
    return out


NO_BN_BATCH = bool(int(__import__("os").environ.get("RL_NO_BN_BATCH", "0")))      # diagnostics: one launch per BatchNorm fold


def bn_finalize(stats, rows: int, tile: int, C: int, gamma, beta, rmean, rvar, nbt, momentum: float,
                eps: float, training: bool, sync: Optional[SyncGroup] = None, nslots: Optional[int] = None,
                folded_bias: Optional[torch.Tensor] = None, defer: Optional[list] = None, pivoted: bool = False,
                pivot: Optional[torch.Tensor] = None):
    """folded_bias: the layer's conv bias when the producing GEMM did NOT add it (rl_bn_finalize in rl_randlanet.h).
    pivoted: the partial sums are shifted around (m - folded_bias), m = `pivot` when given, else the running mean - the vector
    the producer was given, see gemm(pivot=...).
    pivot (training): the layer's pivot vector (Engine.Pv); the fold leaves this batch's mean of y there for the next step.
    defer (a list): the fold is only queued - the returned tensors are filled by `bn_finalize_flush(defer)`, which runs all
    queued folds as ONE launch; the caller flushes before anything reads them."""
    dev = gamma.device
    nslots = H.row_blocks(rows, tile) if nslots is None else nslots
    if training and sync is not None:            # batch statistics of the GLOBAL batch
        stats = _stats_totals(stats, nslots, C)
        if pivoted:
            # the ranks' sums are shifted around EACH RANK's pivot (its running mean - nothing forces the replicas' buffers to
            # be bit-equal): bring them to pivot 0 in double before they are added - S0 = S + n c, Q0 = Q + 2 c S + n c^2 with
            # the fp32 pivot c = running_mean - folded_bias the producer subtracted (exact in double up to 2^-53: the variance
            # keeps what the shift bought) - and finalize the totals as plain sums
            m = pivot if pivot is not None else rmean
            c = (m - folded_bias if folded_bias is not None else m).to(torch.float64)
            S, Q = stats[0, 0].clone(), stats[0, 1].clone()
            stats[0, 0] = S + rows * c
            stats[0, 1] = Q + 2.0 * c * S + rows * c * c
            pivoted = False
        sync.allreduce(stats)
        nslots, rows = 1, sync.global_rows(rows)
    scale = torch.empty(C, dtype=F32, device=dev)
    shift = torch.empty(C, dtype=F32, device=dev)
    mean = torch.empty(C, dtype=F32, device=dev) if training else None
    invstd = torch.empty(C, dtype=F32, device=dev) if training else None
    _dev_check(stats, gamma, beta, rmean, rvar, nbt, folded_bias, pivot)
    if not training:
        pivot = None
    assert folded_bias is None or folded_bias.numel() == C
    if defer is not None and not NO_BN_BATCH:
        it = H.BnFinalizeItem()
        it.stats, it.count, it.gamma, it.beta = H.ptr(stats), rows, H.ptr(gamma), H.ptr(beta)
        it.running_mean, it.running_var, it.num_batches_tracked = H.ptr(rmean), H.ptr(rvar), H.ptr(nbt)
        it.scale, it.shift, it.save_mean, it.save_invstd = scale.data_ptr(), shift.data_ptr(), H.ptr(mean), H.ptr(invstd)
        it.folded_bias, it.nslots, it.C, it.training, it.momentum, it.eps = H.ptr(folded_bias), nslots, C, int(training), momentum, eps
        it.pivoted = int(bool(pivoted and training))
        it.pivot = H.ptr(pivot)
        defer.append((it, stats, scale, shift, mean, invstd))       # (the tensors stay referenced until the launch is issued)
        return scale, shift, mean, invstd
    H.check(H.lib().rl_bn_finalize(H.ptr(stats), nslots, rows, C, H.ptr(gamma), H.ptr(beta),
                                   H.ptr(rmean), H.ptr(rvar), H.ptr(nbt), momentum, eps, int(training),
                                   scale.data_ptr(), shift.data_ptr(), H.ptr(mean), H.ptr(invstd), H.ptr(folded_bias),
                                   int(bool(pivoted and training)), H.ptr(pivot), _st()),
            "rl_bn_finalize")
    return scale, shift, mean, invstd


def bn_finalize_flush(defer: Optional[list]) -> None:
    """The queued BatchNorm folds as one launch (rl_bn_finalize_batch)."""
    if not defer:
        return
    arr = (H.BnFinalizeItem * len(defer))(*[d[0] for d in defer])
    H.check(H.lib().rl_bn_finalize_batch(arr, len(defer), _st()), "rl_bn_finalize_batch")
    defer.clear()


def _bn_bwd_desc(G: torch.Tensor, g_bstride: int, y: Lazy) -> H.BnBwdDesc:
    _dev_check(G, y.raw)
    assert G.dim() == 2 and G.shape[1] == y.raw.shape[1] and g_bstride == y.bstride, \
        "gradient must share the layout of the tensor it belongs to"
    assert G.shape[0] >= (y.B - 1) * y.bstride + y.n
    d = H.BnBwdDesc()
    d.G, d.Y, d.ld, d.bstride = G.data_ptr(), y.raw.data_ptr(), y.raw.shape[1], y.bstride
    d.B, d.n, d.C, d.act, d.slope = y.B, y.n, y.C, y.act, y.slope
    d.scale, d.shift, d.mean, d.invstd = H.ptr(y.scale), H.ptr(y.shift), H.ptr(y.mean), H.ptr(y.invstd)
    return d


def _bn_bwd_finalize(stats, slots: int, rows: int, Cc: int, dgamma, dbeta, coef, sync: Optional[SyncGroup],
                     also: Optional[list] = None) -> None:
    """dgamma / dbeta = this rank's sums; coef = the means the apply pass subtracts - of the GLOBAL batch with `sync`.
    also: queued finalizes of OTHER layers, tuples (stats, slots, rows, C, dgamma, dbeta, coef) whose sums are complete - they
    go out in this launch (rl_bn_bwd_finalize_batch) and the list is cleared."""
    if also and sync is None and not NO_BN_BATCH:
        todo = [(stats, slots, rows, Cc, dgamma, dbeta, coef)] + list(also)
        arr = (H.BnBwdFinalizeItem * len(todo))()
        for it, (s_, n_, r_, c_, dg_, db_, co_) in zip(arr, todo):
            it.stats, it.count, it.dgamma, it.dbeta, it.coef, it.nslots, it.C = s_.data_ptr(), r_, H.ptr(dg_), H.ptr(db_), co_.data_ptr(), n_, c_
        H.check(H.lib().rl_bn_bwd_finalize_batch(arr, len(todo), _st()), "rl_bn_bwd_finalize_batch")
        also.clear()
        return
    if also:
        for (s_, n_, r_, c_, dg_, db_, co_) in also:
            _bn_bwd_finalize(s_, n_, r_, c_, dg_, db_, co_, sync)
        also.clear()
    H.check(H.lib().rl_bn_bwd_finalize(stats.data_ptr(), slots, rows, Cc, H.ptr(dgamma), H.ptr(dbeta), coef.data_ptr(), _st()),
            "rl_bn_bwd_finalize")
    if sync is not None:
        tot = _stats_totals(stats, slots, Cc)
        sync.allreduce(tot)
        H.check(H.lib().rl_bn_bwd_finalize(tot.data_ptr(), 1, sync.global_rows(rows), Cc, None, None, coef.data_ptr(), _st()),
                "rl_bn_bwd_finalize")


def bn_backward(G: torch.Tensor, y: Lazy, dgamma: Optional[torch.Tensor], dbeta: Optional[torch.Tensor],
                training: bool, sync: Optional[SyncGroup] = None, stats: Optional[tuple] = None, also: Optional[list] = None) -> bool:
    """In place: G (gradient w.r.t. the activated value of `y`) becomes the gradient w.r.t. y.raw.
    stats: (partials, nslots) already left by the kernel that produced G (head_bwd) - no reduce sweep over G and Y here.
    also: other layers' queued backward finalizes (see _bn_bwd_finalize) to send out with this layer's; returns True when they
    went out (the list is then empty) - on the paths without a finalize launch of their own they stay queued."""
    d = _bn_bwd_desc(G, y.bstride, y)
    if stats is not None:
        assert training and y.mean is not None and sync is None
        coef = torch.empty(2 * y.C, dtype=F32, device=G.device)
        _bn_bwd_finalize(stats[0], stats[1], y.rows, y.C, dgamma, dbeta, coef, None, also=also)
        d.coef = coef.data_ptr()
        with _rec("bn_bwd_apply", (y.rows, y.C), 12 * y.rows * y.C, 0):
            H.check(H.lib().rl_bn_bwd_apply(C.byref(d), _st()), "rl_bn_bwd_apply")
        return not also
    if (training and y.mean is not None and sync is None and not NO_BN_SMALL and G.data_ptr() % 16 == 0 and y.raw.data_ptr() % 16 == 0
            and H.lib().rl_bn_bwd_fused_supported(y.rows, y.C, y.raw.shape[1])):
        # a small tensor: reduce, finalize and apply in one launch (a workgroup owns a channel quad and all its rows)
        with _rec("bn_bwd_fused", (y.rows, y.C), 12 * y.rows * y.C, 0):
            H.check(H.lib().rl_bn_bwd_fused(C.byref(d), y.rows, H.ptr(dgamma), H.ptr(dbeta), None, _st()), "rl_bn_bwd_fused")
        return not also
    if training and y.mean is not None:
        stats = new_stats(G.device, y.C)
        coef = torch.empty(2 * y.C, dtype=F32, device=G.device)
        d.stats = stats.data_ptr()
        with _rec("bn_bwd_reduce", (y.rows, y.C), 8 * y.rows * y.C, 0):
            H.check(H.lib().rl_bn_bwd_reduce(C.byref(d), _st()), "rl_bn_bwd_reduce")
        _bn_bwd_finalize(stats, H.lib().rl_bn_bwd_slots(y.rows), y.rows, y.C, dgamma, dbeta, coef, sync, also=also)
        d.coef = coef.data_ptr()
    with _rec("bn_bwd_apply", (y.rows, y.C), 12 * y.rows * y.C, 0):
        H.check(H.lib().rl_bn_bwd_apply(C.byref(d), _st()), "rl_bn_bwd_apply")
    return not also


def resid_bn_supported(y1: Lazy, y2: Lazy) -> bool:
    dense = all(y.bstride == y.n and y.raw.shape == (y.B * y.n, y.C) and y.mean is not None and y.act == H.ACT_NONE for y in (y1, y2))
    return dense and y1.C == y2.C and y1.rows == y2.rows and bool(H.lib().rl_resid_bn_bwd_supported(y1.rows, y1.C))


def resid_bn_backward(G: torch.Tensor, O: torch.Tensor, slope: float, y1: Lazy, y2: Lazy, dgamma1, dbeta1, dgamma2, dbeta2,
                      sync: Optional[SyncGroup] = None):
    """Backward of O = LeakyReLU(BN1(y1) + BN2(y2)) down to the two raw tensors: G (dL/dO) becomes the gradient
    w.r.t. y1.raw in place, the returned tensor is the gradient w.r.t. y2.raw (modules.py:325 + two BatchNorm2d)."""
    _dev_check(G, O, y1.raw, y2.raw)
    rows, Cc = y1.rows, y1.C
    assert G.shape == O.shape == (rows, Cc)
    d = H.ResidBnBwdDesc()
    G2 = torch.empty_like(G)
    d.G, d.G2, d.O, d.slope, d.rows, d.C = G.data_ptr(), G2.data_ptr(), O.data_ptr(), slope, rows, Cc
    d.Y1, d.scale1, d.mean1, d.invstd1 = y1.raw.data_ptr(), y1.scale.data_ptr(), y1.mean.data_ptr(), y1.invstd.data_ptr()
    d.Y2, d.scale2, d.mean2, d.invstd2 = y2.raw.data_ptr(), y2.scale.data_ptr(), y2.mean.data_ptr(), y2.invstd.data_ptr()
    if sync is None and not NO_BN_SMALL and H.lib().rl_resid_bn_bwd_fused_supported(rows, Cc):
        with _rec("resid_bn_bwd_fused", (rows, Cc), 24 * rows * Cc, 0):
            H.check(H.lib().rl_resid_bn_bwd_fused(C.byref(d), H.ptr(dgamma1), H.ptr(dbeta1), H.ptr(dgamma2), H.ptr(dbeta2), _st()),
                    "rl_resid_bn_bwd_fused")
        return G2
    st1, st2 = new_stats(G.device, Cc), new_stats(G.device, Cc)
    c1 = torch.empty(2 * Cc, dtype=F32, device=G.device)
    c2 = torch.empty(2 * Cc, dtype=F32, device=G.device)
    d.stats1, d.stats2, d.coef1, d.coef2 = st1.data_ptr(), st2.data_ptr(), c1.data_ptr(), c2.data_ptr()
    with _rec("resid_bn_bwd_reduce", (rows, Cc), 16 * rows * Cc, 0):
        H.check(H.lib().rl_resid_bn_bwd_reduce(C.byref(d), _st()), "rl_resid_bn_bwd_reduce")
    slots = H.lib().rl_bn_bwd_slots(rows)
    if sync is None and not NO_BN_BATCH:
        H.check(H.lib().rl_bn_bwd_finalize_pair(st1.data_ptr(), st2.data_ptr(), slots, rows, Cc, H.ptr(dgamma1), H.ptr(dbeta1), c1.data_ptr(),
                                                H.ptr(dgamma2), H.ptr(dbeta2), c2.data_ptr(), _st()), "rl_bn_bwd_finalize_pair")
    else:
        _bn_bwd_finalize(st1, slots, rows, Cc, dgamma1, dbeta1, c1, sync)
        _bn_bwd_finalize(st2, slots, rows, Cc, dgamma2, dbeta2, c2, sync)
    with _rec("resid_bn_bwd_apply", (rows, Cc), 24 * rows * Cc, 0):
        H.check(H.lib().rl_resid_bn_bwd_apply(C.byref(d), _st()), "rl_resid_bn_bwd_apply")
    return G2


# ------------------------------------------------------------------------------------ rows
def _rows_desc(src: torch.Tensor, src_cols: Tuple[int, int], src_bstride: int, dst: torch.Tensor,
               dst_cols: Tuple[int, int], rows: int, rows_per_batch: int, *, index=None,
               index_shared: bool = False, accumulate: bool = False, lazy: Optional[Lazy] = None):
    _dev_check(src, dst, index)
    d = H.RowsDesc()
    es = src.element_size()
    assert src.dtype == F32 and dst.dtype == F32 and src.dim() == 2 and dst.dim() == 2
    c0s, cn = src_cols
    c0d, cn2 = dst_cols
    assert cn == cn2 and c0s + cn <= src.shape[1] and c0d + cn <= dst.shape[1] and dst.shape[0] >= rows
    d.src, d.lds, d.src_bstride = src.data_ptr() + c0s * es, src.shape[1], src_bstride
    d.dst, d.ldd = dst.data_ptr() + c0d * es, dst.shape[1]
    d.rows, d.rows_per_batch, d.C = rows, rows_per_batch, cn
    if index is not None:
        if index.dtype == torch.int32:
            d.index32 = index.data_ptr()
        else:
            assert index.dtype == torch.int64
            d.index64 = index.data_ptr()
        assert index.numel() >= (rows_per_batch if index_shared else rows)
    d.index_shared, d.accumulate = int(index_shared), int(accumulate)
    if lazy is not None and lazy.scale is not None:
        assert c0s == 0 and cn == lazy.C
        d.scale, d.shift, d.act, d.slope = lazy.scale.data_ptr(), lazy.shift.data_ptr(), lazy.act, lazy.slope
    return d, (8 + 4 * int(accumulate)) * rows * cn


def copy_rows(src: torch.Tensor, src_cols: Tuple[int, int], src_bstride: int, dst: torch.Tensor,
              dst_cols: Tuple[int, int], rows: int, rows_per_batch: int, *, index=None,
              index_shared: bool = False, accumulate: bool = False, lazy: Optional[Lazy] = None) -> None:
    """dst[r, dst_cols] (=|+=) f(src[b*src_bstride + idx, src_cols]); column ranges are (start, count)."""
    d, nbytes = _rows_desc(src, src_cols, src_bstride, dst, dst_cols, rows, rows_per_batch, index=index,
                           index_shared=index_shared, accumulate=accumulate, lazy=lazy)
    with _rec("copy_rows", (rows, src_cols[1], index is not None), nbytes, 0):
        H.check(H.lib().rl_copy_rows(C.byref(d), _st()), "rl_copy_rows")


NO_COPY_PAIR = bool(int(__import__("os").environ.get("RL_NO_COPY_PAIR", "0")))      # diagnostics: one launch per copy


def copy_rows_pair(a: tuple, b: tuple) -> None:
    """Two independent copy_rows in one launch: a, b = (args, kwargs) of copy_rows (the two halves of a concat)."""
    if NO_COPY_PAIR:
        copy_rows(*a[0], **a[1])
        copy_rows(*b[0], **b[1])
        return
    d0, n0 = _rows_desc(*a[0], **a[1])
    d1, n1 = _rows_desc(*b[0], **b[1])
    with _rec("copy_rows_pair", (a[0][5], a[0][1][1], b[0][1][1]), n0 + n1, 0):
        H.check(H.lib().rl_copy_rows_pair(C.byref(d0), C.byref(d1), _st()), "rl_copy_rows_pair")


def scatter_add_rows(src: torch.Tensor, src_cols: Tuple[int, int], dst: torch.Tensor, dst_bstride: int,
                     rows: int, rows_per_batch: int, index: torch.Tensor, index_shared: bool = False) -> None:
    """dst[b*dst_bstride + index[r], :] += src[r, src_cols] (fp32 atomics; dst zeroed by the caller)."""
    _dev_check(src, dst, index)
    d = H.RowsDesc()
    c0s, cn = src_cols
    assert cn == dst.shape[1] and c0s + cn <= src.shape[1] and src.shape[0] >= rows
    d.src, d.lds, d.src_bstride = src.data_ptr() + c0s * 4, src.shape[1], dst_bstride
    d.dst, d.ldd = dst.data_ptr(), dst.shape[1]
    d.rows, d.rows_per_batch, d.C = rows, rows_per_batch, cn
    if index.dtype == torch.int32:
        d.index32 = index.data_ptr()
    else:
        d.index64 = index.data_ptr()
    d.index_shared = int(index_shared)
    with _rec("scatter_add", (rows, cn), 12 * rows * cn, 0):
        H.check(H.lib().rl_scatter_add_rows(C.byref(d), _st()), "rl_scatter_add_rows")


@dataclass
class Csr:
    """Transpose of a neighbour graph idx (B, n_src, k) -> for destination (b, j) the local source rows i*k + kk that
    gathered it, ascending (rl_csr_build)."""
    offsets: torch.Tensor   # (B, n_dst + 1) int32
    entries: torch.Tensor   # (B, n_src*k) int32
    B: int
    n_src: int
    k: int
    n_dst: int


def csr_build(graphs):
    """graphs: list of (idx int32 (B, n_src, k), n_dst).  One launch set for all of them."""
    out = []
    for c0 in range(0, len(graphs), H.CSR_MAX_TASKS):
        chunk = graphs[c0:c0 + H.CSR_MAX_TASKS]
        arr = (H.CsrTask * len(chunk))()
        res = []
        B = chunk[0][0].shape[0]
        nent = 0
        for t, (idx, n_dst) in zip(arr, chunk):
            _dev_check(idx)
            assert idx.dtype == torch.int32 and idx.dim() == 3 and idx.shape[0] == B and n_dst > 0
            _, n_src, k = idx.shape
            off = torch.empty((B, n_dst + 1), dtype=torch.int32, device=idx.device)
            ent = torch.empty((B, n_src * k), dtype=torch.int32, device=idx.device)
            t.idx, t.n_src, t.k, t.n_dst, t.offsets, t.entries = idx.data_ptr(), n_src, k, n_dst, off.data_ptr(), ent.data_ptr()
            res.append(Csr(off, ent, B, n_src, k, n_dst))
            nent += B * n_src * k
        nbytes = H.lib().rl_csr_workspace_bytes(arr, len(chunk), B)
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=chunk[0][0].device)
        with _rec("csr_build", (len(chunk), nent), 16 * nent, 0):
            H.check(H.lib().rl_csr_build(arr, len(chunk), B, ws.data_ptr(), ws.numel(), _st()), "rl_csr_build")
        out.extend(res)
    return out


def segment_sum_rows(src: torch.Tensor, src_cols: Tuple[int, int], src_bstride: int, csr: Csr, dst: torch.Tensor,
                     dst_bstride: int, accumulate: bool = False) -> None:
    """dst[b*dst_bstride + j, :] (=|+=) sum of src[b*src_bstride + r, src_cols] over the rows r that gathered (b, j), in
    ascending r - the gather's backward in a fixed order (no atomics, first writer needs no zero fill)."""
    _dev_check(src, dst, csr.offsets, csr.entries)
    c0, cn = src_cols
    assert src.dtype in (F32, BF16) and dst.dtype == F32 and src.dim() == 2 and dst.dim() == 2
    assert c0 + cn <= src.shape[1] and cn == dst.shape[1]
    assert src.shape[0] >= (csr.B - 1) * src_bstride + csr.n_src * csr.k
    assert dst.shape[0] >= (csr.B - 1) * dst_bstride + csr.n_dst
    d = H.SegsumDesc()
    d.src, d.lds, d.src_bstride = src.data_ptr() + src.element_size() * c0, src.shape[1], src_bstride
    d.src_bf16 = int(src.dtype == BF16)
    d.dst, d.ldd, d.dst_bstride = dst.data_ptr(), dst.shape[1], dst_bstride
    d.offsets, d.entries, d.entries_per_cloud = csr.offsets.data_ptr(), csr.entries.data_ptr(), csr.n_src * csr.k
    d.B, d.n_dst, d.C, d.accumulate = csr.B, csr.n_dst, cn, int(accumulate)
    rows = csr.B * csr.n_src * csr.k
    with _rec("segment_sum", (rows, cn), cn * (src.element_size() * rows + 4 * csr.B * csr.n_dst * (2 if accumulate else 1)) + 4 * rows, 0):
        H.check(H.lib().rl_segment_sum_rows(C.byref(d), _st()), "rl_segment_sum_rows")


# ------------------------------------------------------------------------- pooling, residual
@dataclass
class VirtualRpe:
    """The rpe branch of one encoder level as a function of the coordinates (never stored): stage 1 =
    relu(bn1(rpe . W1^T + b1)), stage 2 = relu(bn2(stage1 . W2^T + b2)) (reference modules.py:313-320).  bn1 / bn2 are
    filled in by the engine once their batch statistics are known: Lazy-like records (scale, shift, mean, invstd)."""
    xyz: torch.Tensor
    idx: torch.Tensor
    d2: torch.Tensor
    B: int
    n: int
    h: int
    W1: torch.Tensor
    b1: torch.Tensor
    W2: torch.Tensor
    b2: torch.Tensor
    bn1: Optional[Lazy] = None
    bn2: Optional[Lazy] = None
    # training: the running means of the two BatchNorms - pivots of the shifted batch statistics (rl_pool_desc.pivot_mean*)
    piv1: Optional[torch.Tensor] = None
    piv2: Optional[torch.Tensor] = None

    @property
    def rows(self) -> int:
        return self.B * self.n * 16


def virtual_rpe_supported(d: int, K: int, points: int = 0, n: int = 0) -> bool:
    """points / n (when given): the kernels of a virtual branch address through 32-bit byte offsets - (points*16) x d/2 row
    tensors below 2 GB, clouds below 2^24 points; larger levels take the stored path."""
    if points * 16 * (d // 2) * 4 >= 2 ** 31 or n >= 2 ** 24:
        return False
    return VIRTUAL_RPE and K == 16 and d in (16, 32, 64) and pool_supported(d, K)


def _fill_virtual(pd: "H.PoolDesc", v: VirtualRpe, stage: int) -> None:
    _dev_check(v.xyz, v.idx, v.d2, v.W1, v.b1, v.W2, v.b2)
    assert v.idx.dtype == torch.int32 and v.idx.shape == (v.B, v.n, 16) and v.d2.shape == v.idx.shape
    assert v.xyz.shape[0] == v.B and v.xyz.shape[1] >= v.n and v.xyz.shape[2] in (3, 4)
    pd.xyz_width = v.xyz.shape[2]
    assert v.W1.numel() == v.h * 10 and v.b1.numel() == v.h and v.W2.numel() == v.h * v.h and v.b2.numel() == v.h
    pd.u_source, pd.xyz, pd.xyz_bstride, pd.nbr_d2 = stage, v.xyz.data_ptr(), v.xyz.shape[1], v.d2.data_ptr()
    pd.W1, pd.b1, pd.W2, pd.b2 = v.W1.data_ptr(), v.b1.data_ptr(), v.W2.data_ptr(), v.b2.data_ptr()
    pd.idx, pd.points, pd.n, pd.d, pd.nbr_k = v.idx.data_ptr(), v.B * v.n, v.n, 2 * v.h, 16
    _dev_check(v.piv1, v.piv2)
    pd.pivot_mean1, pd.pivot_mean2 = H.ptr(v.piv1), H.ptr(v.piv2)
    for tag, bn in (("1", v.bn1), ("2", v.bn2)):
        if bn is not None:
            _dev_check(bn.scale, bn.shift, bn.mean, bn.invstd)
            setattr(pd, f"scale{tag}", bn.scale.data_ptr())
            setattr(pd, f"shift{tag}", bn.shift.data_ptr())
            setattr(pd, f"mean{tag}", H.ptr(bn.mean))
            setattr(pd, f"invstd{tag}", H.ptr(bn.invstd))


def rpe_stats(v: VirtualRpe, stage: int):
    """BatchNorm partial statistics of the raw output of stage 1 / 2 of the virtual rpe branch: (stats, nslots)."""
    pd = H.PoolDesc()
    _fill_virtual(pd, v, stage)
    assert stage == 1 or v.bn1 is not None
    nslots = H.lib().rl_rpe_stats_slots(v.B * v.n)
    stats = torch.empty((nslots, 2, v.h), dtype=torch.float64, device=v.xyz.device)
    with _rec("rpe_stats", (v.rows, v.h, stage), 8 * v.rows, 2 * v.rows * (16 * v.h + (v.h * v.h if stage == 2 else 0))):
        H.check(H.lib().rl_rpe_stats(C.byref(pd), stats.data_ptr(), _st()), "rl_rpe_stats")
    return stats, nslots


def rpe_bn_backward(v: VirtualRpe, stage: int, G: torch.Tensor, dgamma, dbeta, sync: Optional[SyncGroup] = None,
                    stats: Optional[torch.Tensor] = None, nslots: int = 0, queue: Optional[list] = None) -> torch.Tensor:
    """BatchNorm backward statistics of virtual stage `stage` from G (gradient w.r.t. its activated output): fills
    dgamma / dbeta and returns coef (2h floats) for rpe_wgrad.  `stats` / `nslots`: partials already produced by the
    pooling backward kernel that completed G (pool_bwd's bn_bwd_stats) - then no pass over G is made here."""
    if stats is None:
        pd = H.PoolDesc()
        _fill_virtual(pd, v, stage)
        _dev_check(G)
        assert G.shape == (v.rows, v.h) and G.dtype in (F32, BF16)
        pd.rows_bf16 = int(G.dtype == BF16)
        nslots = H.lib().rl_rpe_stats_slots(v.B * v.n)
        stats = torch.empty((nslots, 2, v.h), dtype=torch.float64, device=G.device)
        with _rec("rpe_bn_reduce", (v.rows, v.h, stage), 4 * v.rows * v.h + 8 * v.rows, 0):
            H.check(H.lib().rl_rpe_bn_reduce(C.byref(pd), G.data_ptr(), stats.data_ptr(), _st()), "rl_rpe_bn_reduce")
    coef = torch.empty(2 * v.h, dtype=F32, device=G.device)
    if queue is not None and sync is None:
        queue.append((stats, nslots, v.rows, v.h, dgamma, dbeta, coef))       # goes out with the next layer's finalize launch
    else:
        _bn_bwd_finalize(stats, nslots, v.rows, v.h, dgamma, dbeta, coef, sync)
    return coef


def rpe_wgrad(v: VirtualRpe, stage: int, G: torch.Tensor, coef: torch.Tensor, dW: torch.Tensor, dbias: torch.Tensor,
              pending: list, GU1: Optional[torch.Tensor] = None) -> None:
    """Weight / bias gradient of the stage's Linear (queued on `pending` for the batched slab reduction) and, for stage 2,
    GU1 = the gradient w.r.t. the activated stage-1 output."""
    pd = H.PoolDesc()
    _fill_virtual(pd, v, stage)
    _dev_check(G, coef, dW, dbias, GU1)
    Kin = 10 if stage == 1 else v.h
    assert dW.numel() == v.h * Kin and dbias.numel() == v.h and (stage == 1 or GU1.shape == (v.rows, v.h))
    assert G.dtype in (F32, BF16) and (GU1 is None or GU1.dtype == G.dtype)
    pd.rows_bf16 = int(G.dtype == BF16)
    es = G.element_size()
    floats = H.lib().rl_rpe_wgrad_slab_floats(v.B * v.n, 2 * v.h, stage)
    slab = torch.empty(floats, dtype=F32, device=G.device)
    with _rec("rpe_wgrad", (v.rows, Kin, v.h), es * v.rows * v.h * (2 if stage == 2 else 1) + 8 * v.rows, 2 * v.rows * v.h * (Kin + (v.h if stage == 2 else 0))):
        H.check(H.lib().rl_rpe_wgrad(C.byref(pd), G.data_ptr(), coef.data_ptr(), slab.data_ptr(), slab.numel(), H.ptr(GU1), _st()),
                "rl_rpe_wgrad")
    it = H.WgradReduceItem()
    it.slab, it.dW, it.dbias, it.w_ks, it.w_ns = slab.data_ptr(), dW.data_ptr(), dbias.data_ptr(), 1, Kin
    it.nsplit, it.N, it.K = H.lib().rl_rpe_stats_slots(v.B * v.n), v.h, Kin
    pending.append((it, slab, dW, dbias))


def pool_supported(d: int, K: int) -> bool:
    return (not NO_FUSED_POOL) and bool(H.lib().rl_pool_supported(d, K))


def _pool_desc(u, g: Lazy, idx: torch.Tensor, W: torch.Tensor, n: int, d: int, stage: int = 0) -> H.PoolDesc:
    """u: the rpe-branch half of X - a Lazy tensor ((points*16) x d/2), or a VirtualRpe with `stage` 1 / 2."""
    _dev_check(g.raw, idx, W, g.scale, g.shift)
    h = d // 2
    B = u.B
    assert idx.dtype == torch.int32 and idx.shape == (B, n, 16)
    assert g.raw.shape[1] == h and g.C == h and g.n == n and g.raw.shape[0] >= (B - 1) * g.bstride + n
    assert W.numel() == d * d
    pd = H.PoolDesc()
    if isinstance(u, VirtualRpe):
        assert stage in (1, 2) and u.h == h and u.n == n and u.bn1 is not None and (stage == 1 or u.bn2 is not None)
        _fill_virtual(pd, u, stage)
    else:
        _dev_check(u.raw, u.scale, u.shift)
        assert u.raw.shape == (B * n * 16, h) and u.bstride == u.n == n * 16 and u.C == h
        pd.U, pd.u_scale, pd.u_shift, pd.u_act, pd.u_slope = u.raw.data_ptr(), H.ptr(u.scale), H.ptr(u.shift), u.act, u.slope
    pd.G, pd.g_bstride = g.raw.data_ptr(), g.bstride
    pd.g_scale, pd.g_shift, pd.g_act, pd.g_slope = H.ptr(g.scale), H.ptr(g.shift), g.act, g.slope
    pd.idx, pd.W, pd.points, pd.n, pd.d, pd.nbr_k = idx.data_ptr(), W.data_ptr(), B * n, n, d, 16
    return pd


def pool_fwd(u, g: Lazy, idx: torch.Tensor, W: torch.Tensor, n: int, d: int, stage: int = 0, next_stats: bool = False):
    """Fused gather+concat -> score Linear -> softmax over K -> weighted sum: (B*n, d).  next_stats (virtual stage 1 only):
    also returns the BatchNorm partial statistics of the raw stage-2 output as (pooled, stats, nslots)."""
    pd = _pool_desc(u, g, idx, W, n, d, stage)
    stats2 = None
    if next_stats:
        assert stage == 1
        ns = H.lib().rl_pool_fwd_slots(u.B * n, d)
        stats2 = torch.empty((ns, 2, d // 2), dtype=torch.float64, device=W.device)
        pd.bn_fwd_stats2 = stats2.data_ptr()
    out = torch.empty((u.B * n, d), dtype=F32, device=W.device)
    pd.Pout = out.data_ptr()
    P = u.B * n
    virt = isinstance(u, VirtualRpe)
    # (virtual: index, distance, output + gathered coordinates and rows - DESIGN.md section 5)
    fbytes = P * (4 * (32 + d) + 16 * (16 + 4 * (d // 2))) if virt else 4 * (2 * P * 16 * (d // 2) + P * 16 + P * d)
    with _rec("pool_fwd_virtual" if virt else "pool_fwd", (P, 16, d), fbytes, 2 * P * 16 * d * d):
        H.check(H.lib().rl_pool_fwd(C.byref(pd), _st()), "rl_pool_fwd")
    if next_stats:
        return out, stats2, ns
    return out


def pool_bwd(u, g: Lazy, idx: torch.Tensor, W: torch.Tensor, n: int, d: int, dP: torch.Tensor,
             GU: torch.Tensor, gu_accumulate: bool, dW: torch.Tensor, pending: Optional[list] = None,
             stage: int = 0, bn_bwd_stats: Optional[torch.Tensor] = None, batch: Optional[list] = None) -> torch.Tensor:
    """Backward of the fused pooling block.  Returns DG ((points*16) x d/2): the gradient of the gathered row of every
    neighbourhood slot, to be summed per gathered point with segment_sum_rows.  d <= 64: dW comes out of the kernel.
    d = 128: the kernel writes X and dS and the weight gradient is the ordinary wide kernel on them (queued on `pending`
    like every other layer's)."""
    pd = _pool_desc(u, g, idx, W, n, d, stage)
    _dev_check(dP, GU, dW)
    P = u.B * n
    assert dP.shape == (P, d) and GU.shape == (P * 16, d // 2) and dW.numel() == d * d
    rdt = row_dtype()
    virt = isinstance(u, VirtualRpe)
    assert GU.dtype == (rdt if virt else F32), "GU: the row storage type under a virtual rpe branch, fp32 for a real U tensor"
    pd.rows_bf16 = int(rdt == BF16)
    es = 2 if rdt == BF16 else 4
    DG = torch.empty((P * 16, d // 2), dtype=rdt, device=W.device)
    if bn_bwd_stats is not None:
        _dev_check(bn_bwd_stats)
        assert stage > 0 and bn_bwd_stats.dtype == torch.float64
        assert bn_bwd_stats.numel() >= H.lib().rl_pool_bwd_slots(P, d) * d
        pd.bn_bwd_stats = bn_bwd_stats.data_ptr()
    pd.dP, pd.GU, pd.gu_accumulate, pd.DG, pd.dW = dP.data_ptr(), GU.data_ptr(), int(gu_accumulate), DG.data_ptr(), dW.data_ptr()
    # algorithmic bytes.  Virtual rpe half: per point the forward's 4*(32 + d) (index, distance, output) +
    # 16*(16 + 4*d/2) (gathered coordinates and rows), + 4*d (dP) + DG and GU rows (GU read first when accumulating).  A real U
    # tensor: its rows instead of the coordinates and distances.
    h = d // 2
    if virt:
        nbytes = P * (4 * (32 + d) + 16 * (16 + 4 * h) + 4 * d + es * 16 * h * (2 + int(gu_accumulate)))
    else:
        nbytes = (4 * (2 * P * 16 * h + P * 16 + 2 * P * d) + es * P * 16 * h + 4 * P * 16 * h * (1 + int(gu_accumulate)))
    if d == 128:
        X = torch.empty((P * 16, d), dtype=rdt, device=W.device)
        dS = torch.empty((P * 16, d), dtype=rdt, device=W.device)
        pd.X_out, pd.dS_out = X.data_ptr(), dS.data_ptr()
        with _rec("pool_bwd", (P, 16, d), nbytes + 2 * es * P * 16 * d, 4 * P * 16 * d * d):
            H.check(H.lib().rl_pool_bwd(C.byref(pd), _st()), "rl_pool_bwd")
        wgrad(plain(X, u.B, n * 16), dS, n * 16, d, dW, 1, d, None, pending=pending, batch=batch)
        return DG
    floats = H.lib().rl_pool_slab_floats(P, d)
    if pending is not None:
        # the partial dW slabs join the backward pass's ONE slab reduction (wgrad_flush) instead of a launch of their own
        slab = torch.empty(floats, dtype=F32, device=W.device)
        pd.dW = None
    else:
        slab = _slab(W.device, floats)
    pd.slab, pd.slab_floats = slab.data_ptr(), slab.numel()
    with _rec("pool_bwd", (P, 16, d), nbytes, 6 * P * 16 * d * d):
        H.check(H.lib().rl_pool_bwd(C.byref(pd), _st()), "rl_pool_bwd")
    if pending is not None:
        it = H.WgradReduceItem()
        it.slab, it.dW, it.dbias, it.w_ks, it.w_ns = slab.data_ptr(), dW.data_ptr(), None, 1, d
        it.nsplit, it.N, it.K = H.lib().rl_pool_bwd_grid(P, d, int(virt)), d, d
        pending.append((it, slab, dW, None))
    return DG


def attpool_fwd(X: torch.Tensor, S: torch.Tensor, P: int, K: int) -> torch.Tensor:
    _dev_check(X, S)
    Cc = X.shape[1]
    assert X.shape == S.shape == (P * K, Cc)
    out = torch.empty((P, Cc), dtype=F32, device=X.device)
    with _rec("attpool_fwd", (P, K, Cc), 4 * (2 * P * K * Cc + P * Cc), 0):
        H.check(H.lib().rl_attpool_fwd(X.data_ptr(), S.data_ptr(), P, K, Cc, out.data_ptr(), _st()), "rl_attpool_fwd")
    return out


def attpool_bwd(X, S, Pout, dP, P: int, K: int):
    _dev_check(X, S, Pout, dP)
    Cc = X.shape[1]
    assert X.shape == S.shape == (P * K, Cc) and Pout.shape == dP.shape == (P, Cc)
    dS = torch.empty_like(S)
    dXa = torch.empty_like(X)
    with _rec("attpool_bwd", (P, K, Cc), 4 * (4 * P * K * Cc + 2 * P * Cc), 0):
        H.check(H.lib().rl_attpool_bwd(X.data_ptr(), S.data_ptr(), Pout.data_ptr(), dP.data_ptr(), P, K, Cc,
                                       dS.data_ptr(), dXa.data_ptr(), _st()), "rl_attpool_bwd")
    return dS, dXa


def add_act_fwd(y1: Lazy, y2: Lazy, slope: float) -> torch.Tensor:
    assert y1.rows == y2.rows and y1.C == y2.C and y1.bstride == y1.n and y2.bstride == y2.n
    out = torch.empty((y1.rows, y1.C), dtype=F32, device=y1.raw.device)
    with _rec("add_act", (y1.rows, y1.C), 12 * y1.rows * y1.C, 0):
        H.check(H.lib().rl_add_act_fwd(y1.raw.data_ptr(), y1.scale.data_ptr(), y1.shift.data_ptr(), y2.raw.data_ptr(),
                                       y2.scale.data_ptr(), y2.shift.data_ptr(), y1.rows, y1.C, slope, out.data_ptr(),
                                       _st()), "rl_add_act_fwd")
    return out


def add_act_bwd(G: torch.Tensor, O: torch.Tensor, slope: float) -> None:
    assert G.shape == O.shape and G.is_contiguous() and O.is_contiguous()
    with _rec("add_act", (G.shape[0], G.shape[1]), 12 * G.numel(), 0):
        H.check(H.lib().rl_add_act_bwd(G.data_ptr(), O.data_ptr(), G.shape[0], G.shape[1], slope, _st()), "rl_add_act_bwd")


def scale_mask(x: torch.Tensor, mask: torch.Tensor, scale: float) -> None:
    _dev_check(x, mask)
    assert mask.dtype == torch.uint8 and mask.numel() == x.numel() and x.dtype == F32
    H.check(H.lib().rl_scale_mask(x.data_ptr(), mask.data_ptr(), scale, x.numel(), _st()), "rl_scale_mask")


def dropout_tick(counter: torch.Tensor) -> torch.Tensor:
    """counter[0] += 1 on the device; returns a fresh device scalar holding the new value - the key of one pass's mask."""
    _dev_check(counter)
    assert counter.dtype == torch.int64 and counter.numel() == 1
    key = torch.empty(1, dtype=torch.int64, device=counter.device)
    H.check(H.lib().rl_dropout_tick(counter.data_ptr(), key.data_ptr(), _st()), "rl_dropout_tick")
    return key


def dropout_fwd(x: Lazy, key: torch.Tensor, seed: int, p: float, first_row: int = 0) -> torch.Tensor:
    """Dropout of the (activated) dense tensor x with the Philox mask of (seed, key): (rows, C) output.  first_row: where
    this tensor's row 0 sits in the whole batch's tensor (shards of one batch draw slices of one mask)."""
    _dev_check(x.raw, x.scale, x.shift, key)
    assert x.bstride == x.n and x.raw.shape == (x.rows, x.C) and x.C % 4 == 0
    out = torch.empty_like(x.raw)
    with _rec("dropout", (x.rows, x.C), 8 * x.rows * x.C, 0):
        H.check(H.lib().rl_dropout_fwd(x.raw.data_ptr(), H.ptr(x.scale), H.ptr(x.shift), x.act, x.slope, out.data_ptr(),
                                       x.rows, first_row, x.C, key.data_ptr(), seed & 0xFFFFFFFFFFFFFFFF, p, _st()), "rl_dropout_fwd")
    return out


def dropout_bwd(G: torch.Tensor, key: torch.Tensor, seed: int, p: float, first_row: int = 0) -> None:
    """In place: the gradient through the same mask (regenerated from (seed, key))."""
    _dev_check(G, key)
    assert G.dim() == 2 and G.shape[1] % 4 == 0
    with _rec("dropout", (G.shape[0], G.shape[1]), 8 * G.numel(), 0):
        H.check(H.lib().rl_dropout_bwd(G.data_ptr(), G.shape[0], first_row, G.shape[1], key.data_ptr(), seed & 0xFFFFFFFFFFFFFFFF, p,
                                       _st()), "rl_dropout_bwd")


def upsample_cf(feat: torch.Tensor, idx: torch.Tensor, d2: Optional[torch.Tensor], power: int) -> torch.Tensor:
    """feat (B,F,N1) channel-first, idx/d2 (B,N2,k) -> (B,F,N2)."""
    _dev_check(feat, idx, d2)
    B, Fc, N1 = feat.shape
    _, N2, k = idx.shape
    assert idx.dtype == torch.int32 and feat.dtype == F32 and idx.shape[0] == B
    out = torch.empty((B, Fc, N2), dtype=F32, device=feat.device)
    H.check(H.lib().rl_upsample_cf(feat.data_ptr(), idx.data_ptr(), H.ptr(d2), B, Fc, N1, N2, k, power,
                                   out.data_ptr(), _st()), "rl_upsample_cf")
    return out


def _perm_stride(perm: torch.Tensor, B: int, N: int) -> int:
    """0 for the reference's ONE permutation of a forward, N for one per cloud (band_sort)."""
    assert perm.dtype == torch.int64 and perm.is_contiguous() and perm.numel() in (N, B * N)
    return N if (perm.numel() == B * N and perm.dim() == 2) else 0


def logits_unpermute(lp: torch.Tensor, perm: torch.Tensor, B: int, N: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    Cc = lp.shape[1]
    assert lp.shape == (B * N, Cc)
    if out is None:
        out = torch.empty((B, Cc, N), dtype=F32, device=lp.device)
    else:
        _dev_check(out)
        assert out.shape == (B, Cc, N) and out.dtype == F32
    H.check(H.lib().rl_logits_unpermute_b(lp.data_ptr(), perm.data_ptr(), _perm_stride(perm, B, N), B, N, Cc, out.data_ptr(), _st()),
            "rl_logits_unpermute")
    return out


def logits_permute_grad(dlogits: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    B, Cc, N = dlogits.shape
    _dev_check(dlogits, perm)
    out = torch.empty((B * N, Cc), dtype=F32, device=dlogits.device)
    H.check(H.lib().rl_logits_permute_grad_b(dlogits.data_ptr(), perm.data_ptr(), _perm_stride(perm, B, N), B, N, Cc, out.data_ptr(),
                                             _st()), "rl_logits_permute_grad")
    return out


NO_BAND_SORT = os.environ.get("RL_NO_BAND_SORT") is not None     # A/B: the reference's permutation as drawn
BAND_SORT_MAX_BANDS = 8                                           # (bandsort.hip BS_MAXB: encoder levels + 1)


def band_sort(inp: torch.Tensor, perm: torch.Tensor, edges) -> torch.Tensor:
    """perm (N) -> (B, N): per cloud, the entries of every sampling band [edges[k], edges[k+1]) of the permutation ordered by
    the 4096-cell Morton code of the cloud's points (stable: ties keep the permutation's order).  The sampled SETS are the
    permutation's; the order inside them follows space (rl_band_sort).  inp: (B, N, 3 + F) fp32, x y z first."""
    _dev_check(inp, perm)
    B, N, cin = inp.shape
    assert inp.dtype == F32 and inp.is_contiguous() and perm.dtype == torch.int64 and perm.numel() == N
    e = (C.c_int * len(edges))(*[int(v) for v in edges])
    nb = len(edges) - 1
    need = H.lib().rl_band_sort_workspace_bytes(B, N, e, nb)
    if need < 0:
        H.check(-1, "rl_band_sort_workspace_bytes")
    ws = torch.empty(need, dtype=torch.uint8, device=inp.device)
    out = torch.empty((B, N), dtype=torch.int64, device=inp.device)
    with _rec("band_sort", (B, N), 40 * B * N, 0):
        H.check(H.lib().rl_band_sort(inp.data_ptr(), cin, perm.data_ptr(), B, N, e, nb, out.data_ptr(), ws.data_ptr(), need, _st()),
                "rl_band_sort")
    return out




# ------------------------------------------------------------------------------ loss, adam
LOSS_KINDS = {  # reference trainer.py:244-269
    "cross_entropy": (0, 0.0, 0.0),
    "focal": (1, 0.0, 2.0),
    "dice": (2, 0.5, 1.0),
    "tversky": (2, 0.7, 1.0),
    "focal_tversky": (2, 0.7, 4.0 / 3.0),
}


def loss_forward(logits: torch.Tensor, labels: torch.Tensor, kind: int, alpha: float, gamma: float,
                 neglect_background: bool = True, out: Optional[torch.Tensor] = None, sync: Optional[SyncGroup] = None):
    """Returns (out, work): out[0] = loss, out[1:] metric counts (doubles, on device).  With `sync` the class sums are
    all-reduced first: the loss (and the counts) of the GLOBAL batch, identical on every rank."""
    _dev_check(logits, labels)
    B, Cc, N = logits.shape
    assert labels.shape == (B, N) and labels.dtype == torch.int64 and logits.dtype == F32
    work = torch.empty(H.lib().rl_loss_work_doubles(B * N, Cc), dtype=torch.float64, device=logits.device)
    if out is None:
        out = torch.empty(1 + 4 * Cc, dtype=torch.float64, device=logits.device)
    else:
        _dev_check(out)
        assert out.dtype == torch.float64 and out.numel() == 1 + 4 * Cc
    if sync is not None:
        H.check(H.lib().rl_loss_partials(logits.data_ptr(), labels.data_ptr(), B, Cc, N, kind, gamma, work.data_ptr(), _st()),
                "rl_loss_partials")
        o = H.lib().rl_loss_totals_offset(Cc)
        sync.allreduce(work[o:o + 5 * Cc + 1])
        H.check(H.lib().rl_loss_from_totals(sync.global_rows(B * N), Cc, kind, alpha, gamma, int(neglect_background),
                                            work.data_ptr(), out.data_ptr(), _st()), "rl_loss_from_totals")
        return out, work
    with _rec("loss", (B, Cc, N), 4 * B * Cc * N + 8 * B * N, 0):
        H.check(H.lib().rl_loss_forward(logits.data_ptr(), labels.data_ptr(), B, Cc, N, kind, alpha, gamma,
                                        int(neglect_background), work.data_ptr(), out.data_ptr(), _st()),
                "rl_loss_forward")
    return out, work


def loss_backward(logits, labels, kind: int, alpha: float, gamma: float, neglect_background: bool, work,
                  grad_scale: float = 1.0, sync: Optional[SyncGroup] = None) -> torch.Tensor:
    B, Cc, N = logits.shape
    dlogits = torch.empty_like(logits)
    if sync is not None:
        H.check(H.lib().rl_loss_backward_global(logits.data_ptr(), labels.data_ptr(), B, Cc, N, kind, alpha, gamma,
                                                int(neglect_background), work.data_ptr(), grad_scale, sync.global_rows(B * N),
                                                dlogits.data_ptr(), _st()), "rl_loss_backward_global")
        return dlogits
    with _rec("loss", (B, Cc, N), 8 * B * Cc * N + 8 * B * N, 0):
        H.check(H.lib().rl_loss_backward(logits.data_ptr(), labels.data_ptr(), B, Cc, N, kind, alpha, gamma,
                                         int(neglect_background), work.data_ptr(), grad_scale, dlogits.data_ptr(),
                                         _st()), "rl_loss_backward")
    return dlogits


class Head:
    """What the fused training step hands to Engine.forward so that the network's head - Dropout, fc_end.3, un-permute, loss and
    metric counts - runs as one kernel each way (rl_head_fwd / rl_head_bwd): the labels, the loss, where the step's record goes."""

    def __init__(self, labels: torch.Tensor, kind: int, alpha: float, gamma: float, neglect_background: bool, out: torch.Tensor,
                 grad_scale: float = 1.0):
        self.labels, self.kind, self.alpha, self.gamma, self.neglect = labels, kind, alpha, gamma, neglect_background
        self.out, self.grad_scale = out, grad_scale
        self.work: Optional[torch.Tensor] = None
        self.mask: Optional[torch.Tensor] = None          # the rows' Dropout keep bits, forward -> backward


NO_FUSED_HEAD = bool(int(__import__("os").environ.get("RL_NO_FUSED_HEAD", "0")))      # A/B: the separate launches


def head_supported(x: Lazy, Cc: int) -> bool:
    return (not NO_FUSED_HEAD and x.C == 32 and x.raw.dtype == F32 and x.raw.shape[1] == 32 and x.bstride == x.n
            and bool(H.lib().rl_head_supported(Cc, 32)))


def _head_desc(x: Lazy, W: torch.Tensor, bias: torch.Tensor, perm: torch.Tensor, head: Head, drop) -> "H.HeadDesc":
    _dev_check(x.raw, W, bias, perm, head.labels, head.out, x.scale, x.shift, x.mean, x.invstd)
    B, N, Cc = x.B, x.n, W.shape[0]
    assert W.dtype == F32 and W.numel() == Cc * 32 and bias.numel() == Cc
    assert head.labels.shape == (B, N) and head.labels.dtype == torch.int64 and head.out.numel() == 1 + 4 * Cc
    d = H.HeadDesc()
    d.X, d.scale, d.shift, d.act, d.slope = x.raw.data_ptr(), H.ptr(x.scale), H.ptr(x.shift), x.act, x.slope
    d.mean, d.invstd = H.ptr(x.mean), H.ptr(x.invstd)
    d.W, d.bias, d.perm, d.labels = W.data_ptr(), bias.data_ptr(), perm.data_ptr(), head.labels.data_ptr()
    d.perm_bstride = _perm_stride(perm, x.B, N)
    d.B, d.N, d.C = B, N, Cc
    d.loss_kind, d.alpha, d.gamma, d.neglect_background = head.kind, head.alpha, head.gamma, int(head.neglect)
    key, seed, p_drop, first_row = drop
    d.drop_p, d.drop_key, d.drop_seed, d.drop_first_row = (p_drop if key is not None else 0.0), H.ptr(key), seed, first_row
    d.grad_scale = head.grad_scale
    return d


def head_fwd(x: Lazy, W: torch.Tensor, bias: torch.Tensor, perm: torch.Tensor, head: Head, drop) -> None:
    """Dropout -> fc_end.3 -> un-permute -> loss + counts of the rows of `x` (fc_end.1's lazy output, permuted order) in one
    launch + the loss finalize: fills head.out (rl_loss_forward's record) and head.work.  drop = (key, seed, p, first_row)."""
    d = _head_desc(x, W, bias, perm, head, drop)
    head.work = torch.empty(H.lib().rl_loss_work_doubles(x.rows, d.C), dtype=torch.float64, device=W.device)
    d.work = head.work.data_ptr()
    if d.drop_p > 0.0:
        head.mask = torch.empty(x.rows, dtype=torch.int32, device=W.device)
        d.drop_mask = head.mask.data_ptr()
    with _rec("head_fwd", (x.rows, 32, d.C), 4 * x.rows * 32 + 16 * x.rows, 2 * x.rows * 32 * d.C):
        H.check(H.lib().rl_head_fwd(C.byref(d), head.out.data_ptr(), _st()), "rl_head_fwd")


def head_bwd(x: Lazy, W: torch.Tensor, bias: torch.Tensor, perm: torch.Tensor, head: Head, drop, dW: torch.Tensor,
             dbias: torch.Tensor, pending: list):
    """The whole backward of the head: returns (G, (bn_bwd_partials, nslots)) - G the gradient w.r.t. x's ACTIVATED value, the
    partials what rl_bn_bwd_reduce would leave for x's BatchNorm; fc_end.3's weight / bias gradient slabs join `pending`."""
    d = _head_desc(x, W, bias, perm, head, drop)
    Cc = d.C
    d.work = head.work.data_ptr()
    g = H.lib().rl_head_grid(x.rows)
    G = torch.empty((x.rows, 32), dtype=F32, device=W.device)
    slab = torch.empty(g * (Cc * 32 + Cc), dtype=F32, device=W.device)
    bstats = torch.empty((g, 2, 32), dtype=torch.float64, device=W.device) if x.mean is not None else None
    d.G, d.slab, d.slab_floats, d.bn_bwd_stats = G.data_ptr(), slab.data_ptr(), slab.numel(), H.ptr(bstats)
    d.drop_mask = H.ptr(head.mask)
    with _rec("head_bwd", (x.rows, 32, Cc), 8 * x.rows * 32 + 16 * x.rows, 4 * x.rows * 32 * Cc):
        H.check(H.lib().rl_head_bwd(C.byref(d), _st()), "rl_head_bwd")
    it = H.WgradReduceItem()
    it.slab, it.dW, it.dbias, it.w_ks, it.w_ns = slab.data_ptr(), dW.data_ptr(), dbias.data_ptr(), 1, 32
    it.nsplit, it.N, it.K = g, Cc, 32
    pending.append((it, slab, dW, dbias))
    return G, ((bstats, g) if bstats is not None else None)


def softmax_cf(logits: torch.Tensor) -> torch.Tensor:
    """Softmax over dim 1 of (B,C,N) logits."""
    _dev_check(logits)
    assert logits.dim() == 3 and logits.dtype == F32
    B, Cc, N = logits.shape
    out = torch.empty_like(logits)
    H.check(H.lib().rl_softmax_cf(logits.data_ptr(), B, Cc, N, out.data_ptr(), _st()), "rl_softmax_cf")
    return out


def adam_step(param, grad, exp_avg, exp_avg_sq, lr: torch.Tensor, step: torch.Tensor, beta1=0.9, beta2=0.999,
              eps=1e-8, grad_scale=1.0) -> None:
    _dev_check(param, grad, exp_avg, exp_avg_sq, lr, step)
    n = param.numel()
    assert grad.numel() == n and exp_avg.numel() == n and exp_avg_sq.numel() == n
    assert lr.dtype == F32 and step.dtype == torch.int64
    with _rec("adam", (n,), 28 * n, 0):
        H.check(H.lib().rl_adam_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), n,
                                     lr.data_ptr(), beta1, beta2, eps, grad_scale, step.data_ptr(), _st()),
                "rl_adam_step")
