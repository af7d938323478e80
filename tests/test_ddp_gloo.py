"""The N>1 host path on CPU: world_size-2 `gloo` processes exercise exactly the code bench.py /
TrainStep use around the kernels - flat parameter re-homing, replica broadcast, the single
flat-gradient all-reduce and batch sharding.  (The kernels themselves need the GPU.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(repo, "3d_recognizer_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from randlanet._train import FlatParameters, broadcast_flat, shard_range, sync_gradients
        from randlanet.utils.modules import RandLANet, RandLANetSettings
        torch.manual_seed(100 + rank)                     # replicas start DIFFERENT on purpose
        net = RandLANet(RandLANetSettings(n_classes=3, n_neighbors=8, layer_sizes=[8, 16, 32, 32]),
                        torch.device("cpu"))
        keys_before = list(net.state_dict().keys())
        flat = FlatParameters(net)
        assert list(net.state_dict().keys()) == keys_before
        assert all(p.data_ptr() % 16 == 0 for p in net.parameters())          # dwordx4-loadable weights
        assert all(p.grad is not None and p.grad.data_ptr() % 16 == 0 for p in net.parameters())
        broadcast_flat(flat.param, world)
        ref = [torch.zeros_like(flat.param) for _ in range(world)]
        dist.all_gather(ref, flat.param)
        assert all(torch.equal(ref[0], r) for r in ref), "replicas differ after broadcast"
        # parameters are views: an in-place optimiser update through the module is seen in the buffer
        with torch.no_grad():
            net.fc_start.bias.add_(1.0)
        off = net.fc_start.weight.numel()
        assert torch.allclose(flat.param[off:off + 8], ref[0][off:off + 8] + 1.0)
        # the collective: rank-dependent gradients written through the per-parameter views
        for i, (name, p) in enumerate(net.named_parameters()):
            flat.grads[name].fill_(float((rank + 1) * (i + 1)))
        sync_gradients(flat.grad, world)
        total = sum(r + 1 for r in range(world))
        for i, (name, p) in enumerate(net.named_parameters()):
            assert torch.all(p.grad == total * (i + 1)), name
        # averaged update = what rl_adam_step's grad_scale = 1/world applies
        assert torch.all(net.fc_end[3].conv.bias.grad / world == total * len(list(net.parameters())) / world)
        # sharding: disjoint cover of the global batch
        mine = list(shard_range(37, rank, world))
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        flat_idx = [i for part in gathered for i in part]
        assert sorted(flat_idx) == list(range(37)) and abs(len(gathered[0]) - len(gathered[-1])) <= 1
        # rank-distinct permutation streams (bench.py seeds numpy with 1234 + rank)
        np.random.seed(1234 + rank)
        perm = np.random.permutation(64)
        perms = [None] * world
        dist.all_gather_object(perms, perm.tolist())
        assert perms[0] != perms[1]
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_flat_gradient_allreduce_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(30)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_shard_range_single_process():
    from randlanet._train import shard_range
    assert list(shard_range(8, 0, 1)) == list(range(8))
    parts = [list(shard_range(32, r, 8)) for r in range(8)]
    assert all(len(p) == 4 for p in parts) and sum(parts, []) == list(range(32))
    parts = [list(shard_range(5, r, 4)) for r in range(4)]
    assert [len(p) for p in parts] == [2, 1, 1, 1]
