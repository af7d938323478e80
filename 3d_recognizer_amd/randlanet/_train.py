"""Fused training step of the hot path: forward + loss/metrics + backward + gradient all-reduce +
Adam as one launch schedule on one HIP stream, optionally captured into a hipGraph.

This is the inner loop of Trainer.train (reference randlanet/utils/trainer.py:107-131) with the
host out of the way: parameters, gradients and both Adam moments live in four flat fp32
buffers (the module's nn.Parameters are views into them, so state_dict()/optimisers/checkpoints
keep working); the loss and the metric counts come back as ONE packed record per step
(the reference does 2C+2 `.item()` round trips); with world_size > 1 the flat gradient buffer is
all-reduced once per step over RCCL and divided by the world size inside the Adam kernel.
"""
from typing import Dict, Optional

import numpy as np
import torch

import ctypes as C
import os

from . import _hip as H
from . import _ops as ops

_PERM_BY_KERNEL = os.environ.get("RL_PERM_MEMCPY", "0") != "1"      # A/B: hipMemcpyAsync for the step's permutation
# The coordinate-only part of a step (permutation, neighbour searches, graph transposes: 0.55 of 6.7 ms at bs = 8) as its own graph
# on a second stream, under the previous step's kernels (round 5).  Built, bit-identical - and measured: 6.73 -> 6.78 ms per step.
# The kernel trace shows the preparation running on its own hardware queue beside two or three network kernels, and every one of
# them stretching by what the other takes (grid_query 243 -> 293 - 326 us; 67.3 ms of kernel time in a 59.9 ms window): the
# network's kernels already fill the chip, so OFF by default; RL_PREP_PIPELINE=1 / TrainStep(pipeline=True) turns it on.
# With SEVERAL ranks the preparation of step t + 1 is then ordered behind step t's network graph and runs beside step t's gradient
# all-reduce + Adam - a latency-bound collective on a few CUs under half a millisecond of kernels that need no gradient and no
# weight.  Round 5 made that the multi-rank default on the strength of the argument alone; round 6 measured it with a spin kernel
# of the collective's length standing in for the all-reduce on the one leasable GPU (tools/allreduce_standin.py, DESIGN.md
# section 7) and it is opt-in again for every world size: one schedule for all ranks unless the USER asks for the other, and a
# capture that fails raises on that rank instead of quietly running another schedule than its peers.
_PREP_PIPELINE_ENV = os.environ.get("RL_PREP_PIPELINE", "")
_PREP_PIPELINE = _PREP_PIPELINE_ENV == "1"


def _upload_perm(perm_dev: torch.Tensor, staging: torch.Tensor, N: int) -> None:
    """The step's permutation from its pinned staging slot to the device, on the launch stream.  By a KERNEL of the library
    (pinned host memory is device-addressable: the rows cross PCIe as the kernel's loads) rather than hipMemcpyAsync: the
    runtime's copy runs on another hardware queue, and the replayed graph behind it started 28 us late (6.78 -> 6.73 - 6.77 ms per
    step; RL_PERM_MEMCPY=1 restores the copy)."""
    if _PERM_BY_KERNEL and N % 2 == 0 and N >= 2:
        d = H.RowsDesc()
        d.src, d.lds, d.src_bstride = staging.data_ptr(), 4, N // 2           # int64 x N = float32 x 2N = N/2 rows of 16 bytes
        d.dst, d.ldd = perm_dev.data_ptr(), 4
        d.rows, d.rows_per_batch, d.C = N // 2, N // 2, 4
        H.check(H.lib().rl_copy_rows(C.byref(d), H.stream_ptr()), "rl_copy_rows")
    else:
        perm_dev.copy_(staging, non_blocking=True)


class FlatParameters:
    """Re-homes a module's parameters into one flat fp32 buffer (plus a matching gradient buffer)."""

    def __init__(self, module: torch.nn.Module):
        named = list(module.named_parameters())
        dev = named[0][1].device
        # every parameter starts on a 16-byte boundary so kernels can use dwordx4 loads on weights
        total = sum(-(-p.numel() // 4) * 4 for _, p in named)
        self.param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grads: Dict[str, torch.Tensor] = {}
        o = 0
        for name, p in named:
            n = p.numel()
            self.param[o:o + n].copy_(p.detach().reshape(-1))
            p.data = self.param[o:o + n].view(p.shape)
            g = self.grad[o:o + n].view(p.shape)
            p.grad = g
            self.grads[name] = g
            o += -(-n // 4) * 4
        self.total = total


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous shard of a global batch for one rank (clouds are independent: no data-path
    collective; remainder items go to the lowest ranks)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def broadcast_flat(buf: torch.Tensor, world: int, group=None, src: int = 0) -> None:
    """Replicas start identical: rank `src`'s flat parameter buffer wins."""
    if world > 1:
        import torch.distributed as dist
        dist.broadcast(buf, src, group=group)


def sync_gradients(flat_grad: torch.Tensor, world: int, group=None, always: bool = False) -> None:
    """THE collective of the path: one all-reduce(SUM) over the flat gradient buffer (5.29 MB for
    config A) - RCCL over xGMI on the GPUs, gloo in the CPU tests.  The division by `world`
    is folded into the Adam kernel (grad_scale).  always: issue it for a one-rank group too."""
    if world > 1 or always:
        import torch.distributed as dist
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)


class TrainState:
    """Everything a training run mutates, once per model: the flat parameter / gradient buffers, both
    Adam moments, the step counter and the learning rate (device scalars, so captured graphs read the
    current values).  Several TrainStep objects (one per batch shape) share one TrainState."""

    def __init__(self, module, lr: float = 1e-2, process_group=None, world_size: int = 1):
        self.module = module
        self.dev = module.device
        if self.dev.type != "cuda":
            raise H.HipKernelError("training needs an MI355X (HIP) device")
        self.flat = FlatParameters(module)
        self.engine = module.engine()
        self.world, self.pg = world_size, process_group
        if world_size > 1:
            # ranks train on different clouds: each draws its own Dropout masks (the equivalence mode overrides this
            # per step: there the ranks hold slices of ONE batch and of one mask)
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.engine.drop_stream = dist.get_rank(process_group)
        self.exp_avg = torch.zeros_like(self.flat.param)
        self.exp_avg_sq = torch.zeros_like(self.flat.param)
        self.lr = torch.tensor([lr], dtype=torch.float32, device=self.dev)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=self.dev)

    def set_lr(self, lr: float) -> None:
        self.lr.fill_(float(lr))

    def snapshot(self):
        return (self.flat.param.clone(), self.exp_avg.clone(), self.exp_avg_sq.clone(), self.step_count.clone(),
                {k: v.clone() for k, v in self.module.named_buffers()}, {k: v.clone() for k, v in self.engine.Pv.items()},
                None if self.engine._drop_counter is None else self.engine._drop_counter.clone())

    def restore(self, snap) -> None:
        self.flat.param.copy_(snap[0]); self.exp_avg.copy_(snap[1]); self.exp_avg_sq.copy_(snap[2])
        self.step_count.copy_(snap[3])
        for k, v in self.module.named_buffers():
            v.copy_(snap[4][k])
        for k, v in self.engine.Pv.items():
            v.copy_(snap[5][k])
        if snap[6] is not None and self.engine._drop_counter is not None:
            self.engine._drop_counter.copy_(snap[6])       # (the capture's warm-up passes draw no Dropout masks of the run)


class TrainStep:
    """One data-parallel training step on the HIP kernels, for one batch shape (B, N).

    step(perm) consumes a host permutation (np.random.permutation(N), drawn by the caller from
    the global numpy RNG like the reference, modules.py:571), runs the schedule and leaves the
    packed loss/metric record in `self.out` (device) / `self.out_host` (pinned, asynchronous)."""

    def __init__(self, module, B: int, N: int, loss: str = "dice", lr: float = 1e-2, use_graph: bool = True,
                 process_group=None, world_size: int = 1, state: Optional[TrainState] = None,
                 sync: Optional[ops.SyncGroup] = None, split_schedule: bool = False, pipeline: Optional[bool] = None):
        """split_schedule: run the multi-rank schedule (forward + backward graph, gradient all-reduce, Adam graph) with ONE
        rank too - a one-rank RCCL group then exercises the collective path on a single GPU (tests/test_rccl_gpu.py).
        sync: the data-parallel EQUIVALENCE mode - BatchNorm batch statistics and the loss' class
        sums of the GLOBAL batch (all-reduced), gradients summed instead of averaged: N ranks on shards reproduce the
        single-process step on the whole batch.  Eager launches only (collectives between kernels).
        pipeline (graph mode; default off for every world size, see _PREP_PIPELINE; RL_PREP_PIPELINE=1 turns it on): the part of a step that
        depends on the input rows and the permutation alone (Engine.prepare: permuted rows, all neighbour searches, graph
        transposes) is captured as its OWN graph and replayed on a second stream, so that step t's preparation runs under
        step t - 1's network kernels (the host submits ahead of the GPU).  Two sets of its outputs alternate (and two
        captures of the network graph, one per set): preparation t + 1 never writes what network t still reads.  Same
        kernels, same order per stream: results are bit-identical to the unpipelined step (tests/test_net_gpu.py)."""
        self.state = state if state is not None else TrainState(module, lr, process_group, world_size)
        self.sync = sync
        if sync is not None:
            use_graph = False
            sync.set_shard(B)               # shards may differ in size: global row counts / mask offsets come from here
        st = self.state
        self.module = module
        self.dev = st.dev
        s = module.settings
        self.B, self.N, self.C = B, N, s.n_classes
        self.kind, self.alpha, self.gamma = ops.LOSS_KINDS[loss]
        self.flat, self.engine = st.flat, st.engine
        self.p_drop = float(module.fc_end[2].p)
        self.world, self.pg = st.world, st.pg
        self.split = bool(split_schedule) or self.world > 1
        self.exp_avg, self.exp_avg_sq, self.lr, self.step_count = st.exp_avg, st.exp_avg_sq, st.lr, st.step_count
        # static buffers; all start VALID (the capture pass launches kernels that index with them)
        self.inp = torch.rand((B, N, 3 + s.n_features), dtype=torch.float32, device=self.dev)
        self.labels = torch.zeros((B, N), dtype=torch.int64, device=self.dev)
        self.perm = torch.arange(N, dtype=torch.int64, device=self.dev)
        # pinned staging ring for the per-step permutation: the host may run a few steps ahead of the GPU,
        # so a slot is rewritten only after the async copy that last read it has executed (event per slot)
        self._perm_ring = [torch.empty(N, dtype=torch.int64).pin_memory() for _ in range(4)]
        self._perm_ring_np = [t.numpy() for t in self._perm_ring]     # plain memcpy, no torch CPU thread pool
        self._perm_events = [None] * 4
        self._perm_slot = 0
        self.out = torch.zeros(1 + 4 * self.C, dtype=torch.float64, device=self.dev)
        self.out_host = torch.zeros(1 + 4 * self.C, dtype=torch.float64).pin_memory()
        self.use_graph = use_graph
        self._g_main: Optional[torch.cuda.CUDAGraph] = None
        self._g_adam: Optional[torch.cuda.CUDAGraph] = None
        # pipelined preparation: set k = step % 2 (its own permutation buffer, Engine.Prep, preparation graph, network graph)
        if pipeline is None:
            pipeline = _PREP_PIPELINE
        self.pipeline = bool(use_graph and sync is None and pipeline)
        self._sets: list = []
        self._side: Optional[torch.cuda.Stream] = None
        self._turn = 0
        self._batch_ready: Optional[torch.cuda.Event] = None
        self._net_done: Optional[torch.cuda.Event] = None

    # -- the schedule ------------------------------------------------------------------------
    def _fwd_bwd(self, perm: Optional[torch.Tensor] = None, prep=None):
        self.engine.sync = self.sync
        stream = self.engine.drop_stream
        if self.sync is not None:
            self.engine.drop_stream = 0     # one mask for the whole batch, sliced by sync.cloud_offset
        try:
            # the head of the network (Dropout, fc_end.3, un-permute, loss + counts) as one kernel each way where it is supported
            head = ops.Head(self.labels, self.kind, self.alpha, self.gamma, True, self.out) if self.sync is None else None
            logits, ctx = self.engine.forward(self.inp, self.perm if perm is None else perm, True, self.p_drop, prep=prep, head=head)
            if logits is None:
                self.engine.backward(ctx, None, self.flat.grads)
            else:
                _, work = ops.loss_forward(logits, self.labels, self.kind, self.alpha, self.gamma, True, out=self.out, sync=self.sync)
                dlogits = ops.loss_backward(logits, self.labels, self.kind, self.alpha, self.gamma, True, work, sync=self.sync)
                self.engine.backward(ctx, dlogits, self.flat.grads)
        finally:
            self.engine.sync = None
            self.engine.drop_stream = stream

    def _adam(self):
        # per-rank losses are averaged (grad / world); the equivalence mode's loss is already the global one (grad summed)
        ops.adam_step(self.flat.param, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.lr, self.step_count,
                      grad_scale=1.0 if self.sync is not None else 1.0 / self.world)

    def _allreduce(self):
        if self.sync is not None:
            if self.sync.world > 1:
                self.sync.allreduce(self.flat.grad)
            return
        sync_gradients(self.flat.grad, self.world, self.pg, always=self.split)

    def capture(self, warmup: int = 2) -> None:
        """Run a few eager steps on a side stream (allocator warm-up), then capture."""
        with torch.cuda.device(self.dev):       # launches go to the current device's stream
            self._capture(warmup)

    def _capture(self, warmup: int) -> None:
        self.module.train()
        if not self.use_graph:
            return
        snap = self.state.snapshot()
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._fwd_bwd()
                self._adam()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        # thread_local: RCCL's watchdog thread polls events while we capture; only this thread's calls count
        if self.pipeline:
            self._capture_pipeline()             # (asked for explicitly: a failure raises here, on this rank)
        if not self.pipeline:
            self._g_main = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_main, capture_error_mode="thread_local"):
                self._fwd_bwd()
                if not self.split:
                    self._adam()
        if self.split:
            self._g_adam = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_adam, capture_error_mode="thread_local"):
                self._adam()
        # the warm-up and capture passes must not count as training: restore the state
        self.state.restore(snap)
        torch.cuda.synchronize(self.dev)

    def _capture_pipeline(self) -> None:
        """Two sets {permutation, preparation graph -> Engine.Prep, network graph reading that Prep}.  The preparation graphs are
        captured on the side stream (their outputs live in their own graph pools: static addresses the network graphs bake in)."""
        main = torch.cuda.current_stream(self.dev)
        self._side = torch.cuda.Stream(self.dev)
        self._sets = []
        for k in range(2):
            st = dict(perm=self.perm if k == 0 else self.perm.clone(), prep=None, g_prep=torch.cuda.CUDAGraph(),
                      g_main=torch.cuda.CUDAGraph(), prep_done=None, main_done=None)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                self.engine.prepare(self.inp, st["perm"], True)            # (allocator warm-up of this stream)
                with torch.cuda.graph(st["g_prep"], stream=self._side, capture_error_mode="thread_local"):
                    st["prep"] = self.engine.prepare(self.inp, st["perm"], True)
                st["g_prep"].replay()                                       # valid contents for the capture pass below
            main.wait_stream(self._side)
            # (the two network graphs never run at the same time and keep nothing between steps: one memory pool for both)
            pool = self._sets[0]["g_main"].pool() if k else None
            with torch.cuda.graph(st["g_main"], pool=pool, capture_error_mode="thread_local"):
                self._fwd_bwd(st["perm"], st["prep"])
                if not self.split:
                    self._adam()
            self._sets.append(st)
        self._g_main = self._sets[0]["g_main"]      # (what "a captured step exists" is asked of)

    def set_batch(self, inp: torch.Tensor, labels: torch.Tensor) -> None:
        """The batch of the steps submitted from now on (current stream).  Pipelined mode: these copies sit behind every network
        graph submitted so far - each of which waited for its own preparation - so no preparation still reads the old rows."""
        self.inp.copy_(inp, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)
        if self.pipeline:
            self._batch_ready = torch.cuda.Event()
            self._batch_ready.record(torch.cuda.current_stream(self.dev))

    def step(self, perm: np.ndarray) -> None:
        with torch.cuda.device(self.dev):
            self._step(perm)

    def _step_pipelined(self, perm: np.ndarray) -> None:
        """side stream:  [wait: the batch is there, network graph k of two steps ago is done]  permutation -> set k, preparation graph k
        main stream:  [wait: preparation k]  network graph k  (all-reduce, Adam graph).
        The host runs ahead of the GPU, so preparation k is submitted while network graph 1 - k of the previous step executes."""
        main = torch.cuda.current_stream(self.dev)
        st = self._sets[self._turn]
        self._turn ^= 1
        slot = self._perm_slot
        self._perm_slot = (slot + 1) % len(self._perm_ring)
        if self._perm_events[slot] is not None:
            self._perm_events[slot].synchronize()
        self._perm_ring_np[slot][:] = perm
        # the side stream waits for exactly two things of the main stream: the batch (set_batch's copies) and the last reader of
        # this set, network graph k of two steps ago - NOT for the previous step's network graph, which it is to run beside
        if self._batch_ready is not None:
            self._side.wait_event(self._batch_ready)
        if st["main_done"] is not None:
            self._side.wait_event(st["main_done"])
        # several ranks: the preparation starts when the PREVIOUS step's network graph has finished, i.e. beside that step's
        # gradient all-reduce and Adam - the collective (latency-bound, a few CUs) hides under ~0.5 ms of coordinate-only kernels
        # that need neither gradients nor weights, instead of standing alone between two graphs.  (One rank: started earlier it
        # only time-slices with the network kernels - measured neutral - so the same order costs nothing there.)
        if self.split and self._net_done is not None:
            self._side.wait_event(self._net_done)
        with torch.cuda.stream(self._side):
            _upload_perm(st["perm"], self._perm_ring[slot], self.N)
            ev = torch.cuda.Event()
            ev.record(self._side)
            self._perm_events[slot] = ev
            st["g_prep"].replay()
            done = torch.cuda.Event()
            done.record(self._side)
        main.wait_event(done)
        st["g_main"].replay()
        st["main_done"] = torch.cuda.Event()
        st["main_done"].record(main)
        self._net_done = st["main_done"]
        if self.split:
            self._allreduce()
            self._g_adam.replay()
        self.out_host.copy_(self.out, non_blocking=True)

    def _step(self, perm: np.ndarray) -> None:
        if self.pipeline and self._sets and self._g_main is not None:
            return self._step_pipelined(perm)
        slot = self._perm_slot
        self._perm_slot = (slot + 1) % len(self._perm_ring)
        if self._perm_events[slot] is not None:
            self._perm_events[slot].synchronize()
        staging = self._perm_ring[slot]
        self._perm_ring_np[slot][:] = perm
        _upload_perm(self.perm, staging, self.N)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        self._perm_events[slot] = ev
        if self._g_main is not None:
            self._g_main.replay()
            if self.split:
                self._allreduce()
                self._g_adam.replay()
        else:
            self._fwd_bwd()
            self._allreduce()
            self._adam()
        self.out_host.copy_(self.out, non_blocking=True)     # (by a kernel like the permutation: measured, 0.02 ms slower)

    def last_metrics(self) -> Dict[str, float]:
        """Synchronises and unpacks the record of the last step (reference metrics.py:8-59).  With several
        ranks the counts are summed and the loss averaged over the ranks first."""
        if self.world > 1 and self.sync is None:
            import torch.distributed as dist
            rec_dev = self.out.clone()
            dist.all_reduce(rec_dev, op=dist.ReduceOp.SUM, group=self.pg)
            rec = rec_dev.cpu().numpy()
            rec[0] /= self.world
        else:
            torch.cuda.current_stream(self.dev).synchronize()
            rec = self.out_host.numpy().copy()
        return unpack_record(rec, self.C)


def unpack_record(rec: np.ndarray, C: int) -> Dict[str, float]:
    """Loss and metrics of one step from its packed record [loss, 3 x C class counts, ...] (reference metrics.py:8-59)."""
    from .utils.metrics import accuracy_from_counts, iou_from_counts
    cnt = rec[1:1 + 3 * C].reshape(3, C)
    oa, pca = accuracy_from_counts(cnt)
    miou, pci = iou_from_counts(cnt)
    return dict(loss=float(rec[0]), OA=oa, mAcc=float(np.mean(pca)), mIoU=miou, per_class_iou=pci, per_class_acc=pca)


class RecordTable:
    """The packed records of an epoch's steps, kept on the device: the training loop appends a row per step (one tiny
    device-to-device copy, no synchronisation - the host runs ahead of the GPU like bench.py does) and reads the whole
    table back ONCE at the end of the epoch; with several ranks the table is all-reduced once, not per step."""

    def __init__(self, device, width: int, rows: int = 64):
        self.table = torch.zeros((max(1, rows), width), dtype=torch.float64, device=device)
        self.k = 0

    def append(self, rec: torch.Tensor) -> None:
        if self.k >= self.table.shape[0]:
            self.table = torch.cat([self.table, torch.zeros_like(self.table)])
        self.table[self.k].copy_(rec, non_blocking=True)
        self.k += 1

    def read(self, world: int = 1, group=None) -> np.ndarray:
        """(steps, width) on the host, in step order; counts summed and the loss averaged over the ranks."""
        t = self.table[:self.k]
        if world > 1 and self.k:
            import torch.distributed as dist
            t = t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            t[:, 0] /= world
        host = t.cpu().numpy()
        self.k = 0
        return host


class InferStep:
    """Eval-mode forward for one batch shape (B, N), replayed as one hipGraph: logits (B,C,N) in the original point order
    (RandLANet.forward in eval mode, modules.py:542-611).  The permutation is drawn by the caller like in training."""

    def __init__(self, module, B: int, N: int, use_graph: bool = True):
        self.module = module
        self.dev = module.device
        s = module.settings
        self.engine = module.engine()
        self.inp = torch.rand((B, N, 3 + s.n_features), dtype=torch.float32, device=self.dev)
        self.perm = torch.arange(N, dtype=torch.int64, device=self.dev)
        self.logits = torch.zeros((B, s.n_classes, N), dtype=torch.float32, device=self.dev)
        # pinned staging ring for the permutation (as in TrainStep): the host may run a few passes ahead of the GPU
        self._ring = [torch.empty(N, dtype=torch.int64).pin_memory() for _ in range(4)]
        self._ring_np = [t.numpy() for t in self._ring]
        self._ring_events = [None] * 4
        self._slot = 0
        self.use_graph = use_graph
        self._g: Optional[torch.cuda.CUDAGraph] = None

    def _fwd(self):
        self.engine.forward(self.inp, self.perm, False, logits_out=self.logits)

    def capture(self, warmup: int = 2) -> None:
        with torch.cuda.device(self.dev):
            self._capture(warmup)

    def _capture(self, warmup: int) -> None:
        self.module.eval()
        if not self.use_graph:
            return
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._fwd()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self._g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g, capture_error_mode="thread_local"):
            self._fwd()
        torch.cuda.synchronize(self.dev)

    def step(self, perm: np.ndarray) -> torch.Tensor:
        with torch.cuda.device(self.dev):
            return self._step(perm)

    def _step(self, perm: np.ndarray) -> torch.Tensor:
        slot = self._slot
        self._slot = (slot + 1) % len(self._ring)
        if self._ring_events[slot] is not None:
            self._ring_events[slot].synchronize()    # the slot is free again once the copy that last read it has executed
        self._ring_np[slot][:] = perm
        _upload_perm(self.perm, self._ring[slot], self.perm.numel())
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        self._ring_events[slot] = ev
        if self._g is not None:
            self._g.replay()
        else:
            self._fwd()
        return self.logits
