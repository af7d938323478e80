"""The multi-rank schedule of TrainStep on the one leasable GPU.  (1) RCCL: a ONE-rank `nccl` process group runs it - forward + backward
graph, all-reduce(SUM) of the flat gradient buffer through RCCL on the step's stream, Adam graph, with the NEXT step's
coordinate-only preparation on a second stream beside the collective (and without it) - and must leave exactly the
parameters of the single-graph schedule (a one-rank sum is the identity, grad_scale 1 / 1).  What this covers that the gloo
tests cannot: the RCCL library loads and builds a communicator on this image, graph capture coexists with its watchdog thread
(capture_error_mode "thread_local"), and the collective is ordered between the two graph replays.  Two ranks cannot share a
device under RCCL, so RCCL with more than one rank is the driver's 8-GPU run.  (2) Two and four ranks for real, over gloo
(which moves device tensors through the host): the same schedule - graphs, flat all-reduce, Adam with gradient / world, the
pipelined preparation, the start-up broadcast - with replicas that must stay bit-equal, and equal to the one-rank run when
every rank holds the same batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(n_classes=3, n_points=2048, n_neighbors=16, layer_sizes=[8, 16, 32, 32])
N, B, STEPS = 2048, 2, 5


def _run(split, pg=None, pipeline=None):
    from randlanet._train import TrainStep, broadcast_flat
    from randlanet.utils.modules import RandLANet, RandLANetSettings
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = RandLANet(RandLANetSettings(**CFG), dev)
    net.train()
    rs = np.random.RandomState(3)
    xyz = rs.uniform(0, 1, (B, N, 3)).astype(np.float32)
    lab = np.clip(np.floor(xyz[..., 2] * 3), 0, 2).astype(np.int64)
    st = TrainStep(net, B, N, loss="dice", use_graph=True, process_group=pg, split_schedule=split, pipeline=pipeline)
    # the pipelined preparation (next step's coordinate-only part on a second stream beside the all-reduce) is opt-in (round 6)
    assert st.pipeline == bool(pipeline)
    st.set_batch(torch.from_numpy(xyz).to(dev), torch.from_numpy(lab).to(dev))
    if split:
        import torch.distributed as dist
        dist.broadcast(st.flat.param, 0, group=pg)          # the start-up collective of a multi-rank run
    st.capture()
    assert (st._g_adam is not None) == split
    np.random.seed(9)
    losses = []
    for _ in range(STEPS):
        st.step(np.random.permutation(N))
        losses.append(st.last_metrics()["loss"])
    torch.cuda.synchronize()
    return st.flat.param.detach().cpu().numpy().copy(), losses


def _worker(port, q):
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (repo, os.path.join(repo, "3d_recognizer_amd")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        try:
            t = torch.arange(8, dtype=torch.float32, device="cuda")
            dist.all_reduce(t)                              # communicator is built by the first collective
            torch.cuda.synchronize()
            assert dist.get_backend() == "nccl" and t[7].item() == 7.0
            param, losses = _run(True, pipeline=True)       # pipelined preparation beside the collective (opt-in)
            param2, losses2 = _run(True)                    # the default order: graph, all-reduce, Adam graph
            assert losses2 == losses and np.array_equal(param, param2)
            q.put(("ok", param, losses))
        finally:
            dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((traceback.format_exc(), None, None))


@pytest.mark.timeout(300)
def test_one_rank_rccl_group_runs_the_multi_rank_schedule():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(port, q))
    p.start()
    status, param, losses = q.get(timeout=240)
    p.join(30)
    assert status == "ok", status
    ref_param, ref_losses = _run(False)
    assert losses == ref_losses, (losses, ref_losses)
    assert np.array_equal(param, ref_param), float(np.abs(param - ref_param).max())
    print(f"RCCL one-rank schedule: {STEPS} steps, losses {losses}, parameters bit-equal to the single-graph schedule")


# ---- two ranks for real: gloo instead of RCCL (two RCCL ranks cannot share a device), everything else as on an 8-GPU node --------
def _run2(rank, world, same_data, pipeline=None):
    """The replayed data-parallel schedule with TWO ranks: forward + backward graph, all-reduce of the flat gradient buffer over
    the default process group, Adam graph (gradient / world), the next step's preparation on the second stream beside it."""
    from randlanet._train import TrainStep, broadcast_flat
    from randlanet.utils.modules import RandLANet, RandLANetSettings
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = RandLANet(RandLANetSettings(**CFG), dev)
    if same_data:
        net.fc_end[2].p = 0.0       # (ranks draw DIFFERENT Dropout masks by design - Engine.drop_stream = rank - like the processes of a DDP run)
    net.train()
    rs = np.random.RandomState(3 if same_data else 3 + rank)
    xyz = rs.uniform(0, 1, (B, N, 3)).astype(np.float32)
    lab = np.clip(np.floor(xyz[..., 2] * 3), 0, 2).astype(np.int64)
    st = TrainStep(net, B, N, loss="dice", use_graph=True, world_size=world, pipeline=pipeline)
    assert st.split == (world > 1) and st.pipeline == bool(pipeline)
    st.set_batch(torch.from_numpy(xyz).to(dev), torch.from_numpy(lab).to(dev))
    if world > 1:
        if rank:
            st.flat.param.mul_(1.5)                         # a replica that starts elsewhere ...
        broadcast_flat(st.flat.param, world)                # ... and is brought in line by the start-up broadcast
    st.capture()
    np.random.seed(9)
    losses = []
    for _ in range(STEPS):
        st.step(np.random.permutation(N))
        torch.cuda.synchronize()
        losses.append(float(st.out_host[0]))                # this rank's own loss (last_metrics() would all-reduce it)
    return st.flat.param.detach().cpu().numpy().copy(), losses


def _worker2(rank, world, port, q, same_data, pipeline=None):
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (repo, os.path.join(repo, "3d_recognizer_amd")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            param, losses = _run2(rank, world, same_data, pipeline)
            q.put((rank, "ok", param, losses))
        finally:
            dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc(), None, None))


@pytest.mark.timeout(420)
@pytest.mark.parametrize("same_data,world,pipeline", [(True, 2, None), (True, 2, True), (False, 2, None), (False, 4, True)])
def test_ranks_sharing_one_gpu_run_the_replayed_data_parallel_schedule(same_data, world, pipeline):
    """same_data: both ranks hold the SAME batch - the averaged gradient (g + g) / 2 is g exactly, so the two-rank run must
    leave the parameters of the one-rank run bit for bit.  Otherwise: different batches - the replicas must still agree bit
    for bit after every all-reduce (and not with the one-rank run).  pipeline: the default schedule (graph, all-reduce, Adam
    graph) or, opt-in, the next step's preparation on a second stream beside the collective."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker2, args=(r, world, port, q, same_data, pipeline)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, status, param, losses = q.get(timeout=360)
        assert status == "ok", status
        got[rank] = (param, losses)
    for p in procs:
        p.join(30)
    for r in range(1, world):
        assert np.array_equal(got[0][0], got[r][0]), f"replica {r} drifted away from replica 0"
    ref_param, ref_losses = _run2(0, 1, same_data)
    if same_data:
        assert got[0][1] == ref_losses and got[1][1] == ref_losses, (got[0][1], got[1][1], ref_losses)
        assert np.array_equal(got[0][0], ref_param), float(np.abs(got[0][0] - ref_param).max())
    else:
        assert got[0][1][0] == ref_losses[0] and got[1][1][0] != ref_losses[0]       # rank 0's first batch is the reference's
        assert not np.array_equal(got[0][0], ref_param)
    print(f"{world} gloo ranks on one GPU ({'same' if same_data else 'different'} batches): {STEPS} steps, replicas bit-equal"
          + (", equal to the one-rank run" if same_data else ""))
