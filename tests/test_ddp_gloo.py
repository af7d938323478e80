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


# ---------------------------------------------------------------------------------------------------------------
# Trainer.train with several ranks: the control flow around the step (shards, skipped short batch, BatchNorm-buffer
# averaging, broadcast early-stopping decision) with CPU stand-ins for the HIP step machinery.
def _trainer_worker(rank, world, port, q):
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(repo, "3d_recognizer_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.utils.data import DataLoader, TensorDataset

        from randlanet._train import FlatParameters, sync_gradients
        from randlanet.utils import trainer as T
        from randlanet.utils.metrics import MetricCollector, MetricCollectorBag
        from randlanet.utils.modules import RandLANet, RandLANetSettings
        torch.manual_seed(7 + rank)                        # replicas start DIFFERENT: train() must broadcast
        np.random.seed(5)
        net = RandLANet(RandLANetSettings(n_classes=3, n_neighbors=8, layer_sizes=[8, 16, 32, 32]), torch.device("cpu"))
        steps, evals, clouds_seen = [], [], []

        class State:
            def __init__(self, model, lr, world_):
                self.flat, self.lr, self.world = FlatParameters(model), lr, world_

            def set_lr(self, lr):
                self.lr = lr

        class Step:
            def __init__(self, model, B, N, loss, use_graph, state):
                self.st, self.B, self.model = state, B, model

            def capture(self):
                pass

            def set_batch(self, inp, labels):
                self.inp = inp
                clouds_seen.append([int(round(float(v) * 1000.0 / (64 * 3))) for v in inp[:, 0, 0]])

            def step(self, perm):
                assert self.B > 0, "a rank must never be handed an empty shard"
                self.st.flat.grad.fill_(float(self.inp.sum()))            # shard-dependent gradient
                sync_gradients(self.st.flat.grad, self.st.world)          # THE collective: every rank must arrive
                self.st.flat.param.sub_(self.st.lr * self.st.flat.grad / self.st.world)
                for name, buf in self.model.named_buffers():              # per-shard running statistics
                    if buf.is_floating_point():
                        buf.add_(0.01 * (rank + 1))
                steps.append(self.B)

            def last_metrics(self):
                t = torch.ones(1)
                dist.all_reduce(t)                                        # the real one all-reduces the metric record
                return dict(loss=1.0, OA=0.5, mAcc=0.5, mIoU=0.5, per_class_iou=[0.5] * 3, per_class_acc=[0.5] * 3)

        def fake_evaluate(model, loader, class_names=None, loss_function="dice", postprocess=False, n_evaluations=10):
            # rank 1 would see its metric fall (and stop after `patience` epochs) if it decided alone
            evals.append([float(b.clone().sum()) for n, b in model.named_buffers() if b.is_floating_point()][:3])
            mc = MetricCollector(class_names)
            epoch = len(evals)
            miou = 0.1 * epoch if rank == 0 else 0.9 - 0.1 * epoch
            mc.push(0.5, 0.5, [0.5] * 3, miou, [miou] * 3)
            return MetricCollectorBag([mc], class_names)

        T.Trainer._make_state = staticmethod(lambda model, lr, world_: State(model, lr, world_))
        T.Trainer._make_stepper = staticmethod(lambda model, B, N, loss, g, state: Step(model, B, N, loss, g, state))
        T.Trainer.evaluate = staticmethod(fake_evaluate)
        n = 64
        xyz = torch.arange(9 * n * 3, dtype=torch.float32).view(9, n, 3) / 1000.0
        ds = TensorDataset(xyz, torch.zeros(9, n, dtype=torch.int64), torch.arange(9))
        # batches of 4, 4 and ONE cloud (< world); SHUFFLED, and the ranks' torch streams differ (seed 7 + rank): the
        # shards only partition a batch because train() makes the ranks share one seed per epoch
        loader = DataLoader(ds, batch_size=4, shuffle=True)
        tr = T.Trainer(loader, loader, None, ["a", "b", "c"])
        seen = []
        out = tr.train(net, T.TrainingSettings(epochs=4, batch_size=4, early_stopping=True, early_stopping_patience=2),
                       callbacks=[lambda e, m: seen.append((e, m["val_mIoU"]))])
        assert out is net
        assert steps == [2, 2] * 4, steps                                   # 2 clouds per rank, the 1-cloud batch skipped everywhere
        both_seen = [None] * world
        dist.all_gather_object(both_seen, clouds_seen)
        for step_i, (mine, theirs) in enumerate(zip(*both_seen)):           # every step: 4 DISTINCT clouds over the two ranks
            assert len(set(mine) | set(theirs)) == 4, (step_i, mine, theirs)
        for e in range(4):                                                  # every epoch: 8 distinct clouds trained on
            ids = sum(both_seen[0][2 * e:2 * e + 2] + both_seen[1][2 * e:2 * e + 2], [])
            assert len(set(ids)) == 8, (e, ids)
        # every rank followed rank 0's rising metric: 4 epochs, no early stop, same monitored values
        assert [e for e, _ in seen] == [1, 2, 3, 4] and np.allclose([v for _, v in seen], [0.1, 0.2, 0.3, 0.4])
        # replicas and their BatchNorm buffers are identical on all ranks when validation starts and at the end
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        bufs = torch.cat([b.detach().reshape(-1).double() for _, b in net.named_buffers()])
        for t in (flat, bufs, torch.tensor(evals, dtype=torch.float64)):
            both = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(both, t)
            assert torch.equal(both[0], both[1])
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_trainer_multi_rank_control_flow_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(30)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"
