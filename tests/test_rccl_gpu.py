"""RCCL on the one leasable GPU: a ONE-rank `nccl` process group runs the multi-rank schedule of TrainStep - forward + backward
graph, all-reduce(SUM) of the flat gradient buffer through RCCL on the step's stream, Adam graph, with the NEXT step's
coordinate-only preparation on a second stream beside the collective (and without it) - and must leave exactly the
parameters of the single-graph schedule (a one-rank sum is the identity, grad_scale 1 / 1).  What this covers that the gloo
tests cannot: the RCCL library loads and builds a communicator on this image, graph capture coexists with its watchdog thread
(capture_error_mode "thread_local"), and the collective is ordered between the two graph replays.  Two ranks cannot share a
device under RCCL, so more than one rank is the driver's 8-GPU run."""
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
    # several ranks (or their schedule): the next step's coordinate-only preparation runs on a second stream beside the all-reduce
    assert st.pipeline == (split if pipeline is None else pipeline)
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
            param, losses = _run(True)                      # (pipelined preparation: the multi-rank default)
            param2, losses2 = _run(True, pipeline=False)    # the plain order: graph, all-reduce, Adam graph
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
