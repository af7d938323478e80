"""Training / evaluation loops behind Model.train and Model.evaluate (reference
randlanet/utils/trainer.py:23-367): Adam + StepLR(10, gamma), dice loss by default, per-epoch
validation over 10 seeded passes, early stopping on val_mIoU, optional TensorBoard scalars.

What differs from the reference is only where the work happens: the forward/backward is the HIP
launch schedule of RandLANet, the loss is the fused HIP loss, and accuracy + IoU of a batch come
from ONE packed device->host read instead of 2C+2 `.item()` calls (trainer.py:121-131).
"""
import logging
import os
from collections import OrderedDict
from contextlib import contextmanager
from dataclasses import dataclass
from pathlib import Path
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
from torch.utils.data import DataLoader
from tqdm import tqdm

from .early_stopper import EarlyStopper
from .losses import get_loss
from .metrics import (MetricCollector, MetricCollectorBag, accuracy_from_counts, class_counts,
                      iou_from_counts)
from .modules import RandLANet, UpSampler

logger = logging.getLogger("trainer")
logger.setLevel(logging.DEBUG)


@dataclass
class TrainingSettings:
    #: Number of epochs to train
    epochs: int = 150
    #: Size of minibatches used during training
    batch_size: int = 8
    #: Base learning rate
    learning_rate: float = 1e-2
    #: Decay factor applied to the learning rate every 10 epochs
    learning_rate_decay: float = 0.9
    #: "cross_entropy", "focal", "dice", "tversky" or "focal_tversky"
    loss_function: str = "dice"
    #: Early stopping
    early_stopping: bool = True
    #: Patience for early stopping
    early_stopping_patience: int = 20


def _summary_writer(log_dir: Optional[Path]):
    if log_dir is None:
        return None
    try:
        from torch.utils.tensorboard import SummaryWriter
    except Exception:  # tensorboard is an optional dependency of the host glue
        logger.warning("tensorboard is not installed: scalars are not written")
        return None
    return SummaryWriter(str(log_dir))


def _batch_metrics(logits: torch.Tensor, labels: torch.Tensor):
    cnt = class_counts(logits, labels)                 # one launch, one read-back
    oa, pca = accuracy_from_counts(cnt)
    miou, pci = iou_from_counts(cnt)
    return oa, pca, miou, pci


class Trainer:
    def __init__(self, train_dataloader: DataLoader, validation_dataloader: DataLoader,
                 log_dir: Optional[Path] = None, class_names: Optional[List[str]] = None):
        self._train_dataloader = train_dataloader
        self._validation_dataloader = validation_dataloader
        self._log_dir = log_dir
        self._class_names = class_names

    _get_loss = staticmethod(get_loss)

    # The step machinery behind train(); class attributes so that the multi-rank control flow can be exercised on CPU
    # with stand-ins (tests/test_ddp_gloo.py).  The real ones need the MI355X.
    @staticmethod
    def _make_state(model, lr, world):
        from .._train import TrainState
        return TrainState(model, lr, None, world)

    @staticmethod
    def _make_stepper(model, B, N, loss, use_graph, state):
        from .._train import TrainStep
        return TrainStep(model, B, N, loss=loss, use_graph=use_graph, state=state)

    def train(self, model: RandLANet, settings: TrainingSettings,
              callbacks: List[Callable[[int, Dict[str, float]], None]] = []) -> RandLANet:
        """Reference semantics (trainer.py:62-168): Adam(lr) + StepLR(10, decay), per-batch forward / loss /
        backward / step / metrics, per-epoch validation, early stopping on val_mIoU, best weights returned.
        The inner step runs as the fused HIP schedule (`_train.TrainStep`): full batches replay a captured
        hipGraph, a ragged last batch runs the same schedule eagerly; parameters, gradients and Adam moments
        live in flat buffers shared by both.

        With torch.distributed initialised every rank trains on its shard of every batch (the ranks share one torch /
        numpy seed per epoch, broadcast by rank 0, so they all see the same batches) and the gradients are
        all-reduced once per step.  The replicas are kept in lock-step: rank 0's parameters and BatchNorm buffers
        are broadcast before the first step; a batch with fewer clouds than ranks is skipped on ALL ranks (a rank
        without clouds would miss the gradient all-reduce the others wait in); before every validation the
        BatchNorm running statistics - which each rank updates from its own shard - are averaged over the ranks,
        so every rank validates the same model; the early-stopping decision and the choice of the best weights
        follow rank 0's monitored metric on every rank."""
        from .._train import broadcast_flat, shard_range
        world, rank = 1, 0
        dist = torch.distributed
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(), dist.get_rank()
        state = self._make_state(model, settings.learning_rate, world)
        if world > 1:
            broadcast_flat(state.flat.param, world)
            for _, buf in model.named_buffers():
                dist.broadcast(buf, 0)
        steppers: Dict[tuple, object] = {}
        patience = settings.early_stopping_patience if settings.early_stopping else settings.epochs
        stopper = EarlyStopper(patience=patience, metric="val_mIoU")
        model.train()
        logger.info(f"Training on {len(self._train_dataloader.dataset)} training samples and "
                    f"{len(self._validation_dataloader.dataset)} validation samples.")
        writer = _summary_writer(self._log_dir) if rank == 0 else None
        full_batch = self._train_dataloader.batch_size
        lr = settings.learning_rate
        warned_short = False
        table = None                 # the epoch's step records, on the device (RecordTable)
        for epoch in range(1, settings.epochs + 1):
            collected = MetricCollector(self._class_names)
            if world > 1:
                # every rank iterates the SAME loader and keeps its slice of each batch: that only partitions the data if
                # all ranks draw the same shuffle / sampling / augmentation, i.e. share the torch and numpy streams -
                # rank 0 draws a seed per epoch and everybody adopts it
                seed = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64)
                seed_dev = seed.to(model.device)
                dist.broadcast(seed_dev, 0)
                torch.manual_seed(int(seed_dev.item()))
                np.random.seed(int(seed_dev.item()) & 0x7FFFFFFF)
            for batch, labels, _ in tqdm(self._train_dataloader, desc="Training", leave=False, disable=rank != 0):
                if world > 1:           # clouds are independent: each rank takes its contiguous shard
                    if batch.shape[0] < world:
                        if not warned_short and rank == 0:
                            logger.warning(f"A batch of {batch.shape[0]} clouds cannot be shared by {world} ranks: skipped.")
                        warned_short = True
                        continue
                    part = shard_range(batch.shape[0], rank, world)
                    batch, labels = batch[part.start:part.stop], labels[part.start:part.stop]
                key = (batch.shape[0], batch.shape[1])
                if key not in steppers:
                    steppers[key] = self._make_stepper(model, key[0], key[1], settings.loss_function,
                                                       (key[0] == full_batch or world > 1), state)
                    steppers[key].capture()
                    model.train()
                stepper = steppers[key]
                stepper.set_batch(batch.to(model.device, torch.float32), labels.to(model.device))
                stepper.step(np.random.permutation(key[1]))      # the forward's permutation (modules.py:571)
                if getattr(stepper, "out", None) is not None and getattr(stepper, "sync", None) is None:
                    # the step's packed record stays on the device: no synchronisation here, the host runs ahead of the GPU
                    # (the pinned permutation ring of the stepper bounds how far); read back once per epoch, below
                    if table is None:
                        from .._train import RecordTable
                        table = RecordTable(model.device, stepper.out.numel(), len(self._train_dataloader) + 1)
                    table.append(stepper.out)
                else:                                            # (stand-in steppers of the CPU tests, the equivalence mode)
                    m = stepper.last_metrics()
                    collected.push(m["loss"], m["OA"], m["per_class_acc"], m["mIoU"], m["per_class_iou"])
            if table is not None and table.k:
                # ONE read-back (and, with ranks, ONE all-reduce) per epoch; the metrics are pushed in step order, exactly
                # what the per-step loop pushed (reference trainer.py:121-131)
                from .._train import unpack_record
                for rec in table.read(world, getattr(state, "pg", None)):
                    m = unpack_record(rec, model.settings.n_classes)
                    collected.push(m["loss"], m["OA"], m["per_class_acc"], m["mIoU"], m["per_class_iou"])
            if epoch % 10 == 0:                                  # StepLR(step_size=10, gamma) (trainer.py:81-83)
                lr *= settings.learning_rate_decay
                state.set_lr(lr)
            if world > 1:               # every rank validates the same model: average the per-shard running statistics
                for _, buf in model.named_buffers():
                    if buf.is_floating_point():
                        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
                        buf.div_(world)
            validation = Trainer.evaluate(model, self._validation_dataloader, class_names=self._class_names,
                                          loss_function=settings.loss_function)
            metrics = collected.as_dict()
            metrics.update(validation.as_dict("val"))
            if world > 1:               # one decision for all ranks: rank 0's monitored value
                monitored = torch.tensor([float(metrics["val_mIoU"])], dtype=torch.float64, device=model.device)
                dist.broadcast(monitored, 0)
                metrics["val_mIoU"] = float(monitored.item())
            keep_going = stopper.check(metrics, model)
            if rank == 0:
                self._log(epoch, settings.epochs, lr, collected.as_dict(), validation.as_dict(include_stdev=True), writer)
            for callback in callbacks:
                callback(epoch, metrics)
            if not keep_going:
                break
        if writer is not None:
            writer.close()
        best = stopper.load_best_model_weights(model)
        if best is None:
            logger.warning("Model did not improve during training!")
            best = model
        model.eval()
        best.eval()
        return best

    @staticmethod
    def _evaluate_on_device(model: RandLANet, data_loader, class_names, loss_function: str,
                            n_evaluations: int) -> MetricCollectorBag:
        """The seeded passes of evaluate() with the host out of the loop: every batch replays the eval forward's hipGraph
        (`_train.InferStep`, one per batch shape, kept on the model), loss AND class counts come from ONE rl_loss_forward
        launch into a row of a device table, and the whole table is read back ONCE after the last pass (the reference
        does 2C+3 `.item()` round trips per batch, trainer.py:324-354).  Same numbers as the per-batch path: the same
        kernels in the same order, the same permutation draws from the numpy stream seeded 100*i."""
        from .. import _ops as ops
        device = model.device
        C = model.settings.n_classes
        kind, alpha, gamma = ops.LOSS_KINDS[loss_function]
        neglect = kind == 2                                    # FocalTverskyLoss(neglect_background=True), trainer.py:253-267
        n_batches = len(data_loader)
        table = torch.zeros((max(1, n_evaluations * n_batches), 1 + 4 * C), dtype=torch.float64, device=device)
        rows_per_pass: List[int] = []
        k = 0
        with torch.cuda.device(device):
            for i in range(n_evaluations):
                np.random.seed(100 * i)
                first = k
                for batch, labels, _ in tqdm(data_loader, desc="Evaluation", leave=False):
                    if k >= table.shape[0]:                    # a loader without a reliable len(): grow
                        table = torch.cat([table, torch.zeros_like(table)])
                    step = model.infer_step(batch.shape[0], batch.shape[1])
                    step.inp.copy_(batch.to(device, torch.float32), non_blocking=True)
                    logits = step.step(np.random.permutation(batch.shape[1]))
                    ops.loss_forward(logits, labels.to(device, torch.int64).contiguous(), kind, alpha, gamma, neglect,
                                     out=table[k])
                    k += 1
                rows_per_pass.append(k - first)
            host = table[:k].cpu().numpy()                     # THE read-back
        passes: List[MetricCollector] = []
        k = 0
        for n_rows in rows_per_pass:
            current = MetricCollector()
            for rec in host[k:k + n_rows]:
                cnt = rec[1:1 + 3 * C].reshape(3, C)
                oa, pca = accuracy_from_counts(cnt)
                miou, pci = iou_from_counts(cnt)
                current.push(float(np.float32(rec[0])), oa, pca, miou, pci)      # the criterion returns a float32 scalar
            k += n_rows
            passes.append(current)
        return MetricCollectorBag(passes, class_names)

    def _log(self, epoch: int, total_epochs: int, lr: float, train_metrics: OrderedDict,
             validation_metrics: OrderedDict, writer) -> None:
        parts = [f"Epoch {epoch:3d}/{total_epochs:3d}"]
        v = validation_metrics
        parts.append("loss: %.4f - val_loss: %.4f (s: %.4f)" % (train_metrics["loss"], v["loss"][0], v["loss"][1]))
        for key in ("mAcc", "mIoU"):
            parts.append("%s: %.2f%% - val_%s: %.2f%% (s: %.2f%%)" % (
                key, train_metrics[key] * 100, key, v[key][0] * 100, v[key][1] * 100))
        logger.info(" - ".join(parts))
        for mode, metrics in (("Training", train_metrics), ("Validation", validation_metrics)):
            cells = []
            for key in [k for k in metrics if k.endswith(" IoU")]:
                value = metrics[key]
                name = key[: -len(" IoU")]
                if isinstance(value, tuple):
                    cells.append("%s: %5.2f%% (s: %5.2f%%)" % (name, value[0] * 100, value[1] * 100))
                else:
                    cells.append("%s: %5.2f%% %11s" % (name, value * 100, ""))
            logger.info(f"{'':15s} {mode + ' IoU:':16s}" + " - ".join(cells))
        if writer is not None:
            writer.add_scalar("Learning rate", lr, epoch)
            for mode, metrics in (("Train", train_metrics), ("Validation", validation_metrics)):
                for key, value in metrics.items():
                    writer.add_scalar(f"{key}/{mode}", value[0] if isinstance(value, tuple) else value, epoch)

    @staticmethod
    def evaluate(model: RandLANet, data_loader: DataLoader, class_names: Optional[List[str]] = None,
                 loss_function: str = "dice", postprocess: bool = False,
                 n_evaluations: int = 10) -> MetricCollectorBag:
        """n_evaluations passes with numpy seeds 0, 100, 200, ... (the forward's permutation is the only
        randomness in eval mode); the caller's numpy RNG state is restored (trainer.py:271-367)."""

        @contextmanager
        def eval_mode(m: torch.nn.Module):
            was_training = m.training
            m.eval()
            try:
                yield
            finally:
                m.train(was_training)

        criterion = get_loss(loss_function)
        device = model.device
        saved_rng = np.random.get_state()
        if postprocess:
            assert data_loader.batch_size == 1, "Batch size 1 required when evaluating with postprocessing!"
        if device.type == "cuda" and not postprocess and not int(os.environ.get("RL_EVAL_EAGER", "0")):
            with eval_mode(model), torch.no_grad():
                bag = Trainer._evaluate_on_device(model, data_loader, class_names, loss_function, n_evaluations)
            np.random.set_state(saved_rng)
            return bag
        upsampler = UpSampler("nni", device)
        passes: List[MetricCollector] = []
        with eval_mode(model), torch.no_grad():
            for i in range(n_evaluations):
                np.random.seed(100 * i)
                current = MetricCollector()
                for batch, labels, indices in tqdm(data_loader, desc="Evaluation", leave=False):
                    batch, labels = batch.to(device), labels.to(device)
                    logits = model(batch)
                    loss = criterion(logits, labels).item()
                    if postprocess:
                        full_input, full_labels, _ = data_loader.dataset.__getitem__(int(indices[0]), preprocess=False)
                        from .. import _ops as ops
                        conf = ops.softmax_cf(logits.contiguous())
                        scores = upsampler(conf.unsqueeze(-1), batch[:, :, :3],
                                           full_input[:, :3].unsqueeze(0)).squeeze(-1)
                        target = full_labels.unsqueeze(0).to(device)
                    else:
                        scores, target = logits, labels
                    current.push(loss, *_batch_metrics(scores, target))
                passes.append(current)
        np.random.set_state(saved_rng)
        return MetricCollectorBag(passes, class_names)
