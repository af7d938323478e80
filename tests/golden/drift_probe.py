#!/usr/bin/env python3
"""Why the HIP path's validation mIoU sits ~0.02 above the reference's over 128 seeds while the training loss agrees:
an experiment ON THE REFERENCE, in the build container (imports /root/reference like make_golden.py; stores nothing of it).

Hypothesis.  A conv bias in front of a BatchNorm has a true gradient of exactly 0 (it cancels in y - mean).  What the
reference's fp32 autograd leaves there is rounding noise (~1e-5 at these sizes), far above Adam's eps = 1e-8, so Adam
turns it into full +-lr steps of random sign: those biases random-walk.  Training never sees it (batch statistics
re-centre), but in eval mode the running mean lags the last step's move, so every such channel is shifted by +-lr - noise
on the validation forward.  The HIP path accumulates those sums in a fixed order with fp64 statistics and lands at ~0
(below eps): its biases barely move.

Test.  The reference trainer with the gradients of exactly those biases set to 0 before every Adam step, same 64 seeds as
train_seeds.npz; paired difference against the unmodified reference runs stored there.  If the hypothesis holds the
"de-noised" reference moves up by what separates the HIP path from the reference (+0.02 val mIoU, -0.02 val loss) and
its training loss does not move.

Usage:  python tests/golden/drift_probe.py [n_seeds]     -> prints the paired differences per epoch and writes
        train_seeds_denoised.npz (the de-noised reference's histories; tests/test_model_gpu.py brackets the HIP path
        between the reference and this)

Result (64 seeds): training loss unchanged (|d| <= 3e-4 at every epoch); validation loss -0.027 ... -0.044 (2.7 - 4.8 sigma),
validation mIoU +0.02 ... +0.04 (up to 4.2 sigma).  The HIP path sits between the two (+0.02 over 128 seeds).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (imports the reference with the same placeholders)

from randlanet import AugmentationSettings, Model, RandLANetSettings, TrainingSettings  # noqa: E402


class DenoisedAdam(torch.optim.Adam):
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group["params"]:
                if getattr(p, "_true_gradient_is_zero", False) and p.grad is not None:
                    p.grad.zero_()
        return super().step(closure)


def main(n_seeds=64):
    z = np.load(os.path.join(HERE, "train_run.npz"))
    ref = np.load(os.path.join(HERE, "train_seeds.npz"))["histories"][:n_seeds]
    clouds = [(xyz, np.zeros((xyz.shape[0], 0), np.float32), lab.astype(np.int64)) for xyz, lab in zip(z["clouds"], z["labels"])]
    train, val = clouds[:8], clouds[8:]
    import randlanet.utils.trainer as T
    T.torch.optim.Adam = DenoisedAdam            # trainer.py:78 looks Adam up at call time
    hists = []
    for seed in range(n_seeds):
        torch.manual_seed(seed)
        np.random.seed(seed)
        s = RandLANetSettings(n_classes=3, n_points=1024, n_neighbors=16, layer_sizes=[8, 16, 32, 32], knn="approximate")
        model = Model(s, use_gpu=False)
        model.module.fc_end[2].p = 0.0
        for name, p in model.module.named_parameters():
            if name.endswith("conv.bias") and not name.startswith("fc_end.3"):
                p._true_gradient_is_zero = True
        hist = []
        ts = TrainingSettings(epochs=6, batch_size=4, learning_rate=1e-2, early_stopping=False)
        model.train(train, val, ts, AugmentationSettings(), None, ["bg", "a", "b"],
                    callbacks=[lambda e, m: hist.append([m["loss"], m["mIoU"], m["val_loss"], m["val_mIoU"]])])
        hists.append(hist)
        print(f"seed {seed}: val_mIoU {np.round(np.array(hist)[:, 3], 4).tolist()}", flush=True)
    h = np.array(hists)
    np.savez_compressed(os.path.join(HERE, "train_seeds_denoised.npz"), seeds=np.arange(n_seeds), histories=h)
    for col, what in enumerate(("train loss", "train mIoU", "val loss", "val mIoU")):
        d = h[:, :, col] - ref[:, :, col]
        print(f"de-noised reference - reference, per epoch, {what:10s}: " + "  ".join(
            f"{d[:, e].mean():+.4f} ({d[:, e].mean() / (d[:, e].std(ddof=1) / np.sqrt(len(d)) + 1e-30):+.1f}s)" for e in range(d.shape[1])))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 64)
