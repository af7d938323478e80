#!/usr/bin/env python3
"""CPU bisect of the de-noised training-run residual: experiments ON THE REFERENCE, in the build
container (imports /root/reference through make_golden.py; stores numbers only).

Protocol = tests/golden/drift_probe.py (the G6 run: 8 training / 4 validation mock sub-samples, dice, Adam 1e-2, batch 4,
6 epochs, Dropout off, conv biases in front of a BatchNorm frozen), per (torch, numpy) seed pair.  A VARIANT switches ONE
semantic property of the HIP path into the reference as a CPU restatement; the paired difference of the validation mIoU
against the unmodified de-noised reference over the same seeds says whether that property moves the metric.

  base         the de-noised reference itself (another summation order than train_seeds_denoised.npz when run with another
               thread count: the NULL distribution of a paired difference between two draws of one and the same algorithm)
  fcstart      fc_start.bias frozen too (Linear bias in front of bn_start: true gradient 0; the HIP path's fixed-order sums
               leave ~1e-8 there, the reference's autograd ~1e-6 -> +-lr Adam steps)
  bn_sumsq     train-mode BatchNorm statistics as the HIP path formed them up to round 4: var = E[y^2] - E[y]^2 from fp32
               per-lane partial sums (256 interleaved lanes, sequential fp32 accumulation), combined in fp64
  bn_shift     the round-5 form: sums of (y - c), (y - c)^2 around c = the running mean, same partial-sum structure
  adam_eps0    gradients below 1e-7 of a tensor's max flushed to 0 before Adam (what fixed-order fp64 sums do to the
               reference's rounding-noise gradients)
  nobias_fold  conv biases in front of a BatchNorm left out of the tensor, running mean kept as mean + bias

Usage:  python tests/golden/bisect_probe.py --variant base --seeds 0:256 [--procs 8] [--threads 1]
        python tests/golden/bisect_probe.py --report          (table of every stored variant against `base`)
Results: tests/golden/bisect/<variant>.npz (seeds, histories (S, 6, 4): loss, mIoU, val_loss, val_mIoU); the fixtures of
tests/test_model_gpu.py::test_denoised_training_matches_denoised_reference: --export (256 seeds, two draws of `base`) and
--variant fcstart --seeds 0:4096 --fixture train_seeds_denoised_fc4096.npz (one draw; ~25 minutes on 8 cores).
Round 6 added seeds 4096:10240 and 10240:16384 the same way (train_seeds_denoised_fc10240.npz / _fc16384.npz, --procs 7: 64 seeds
per minute) for the 16384-seed comparison of profiles/r06_denoised_16384.txt (tools/denoised_compare.py).
"""
import argparse
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "bisect")


def _patch_batchnorm(mode):
    """Train-mode BatchNorm2d with the batch statistics formed the way the HIP kernels form them (forward value only: the
    backward is autograd's through the same expression, i.e. the exact derivative - as in the kernels, which use the saved
    mean / invstd)."""
    import torch
    import torch.nn.functional as F

    LANES = 256

    def lane_sums(y2d):
        """(rows, C) fp32 -> per-channel sum and sum of squares: LANES interleaved lanes accumulate sequentially in fp32
        (cumsum along the lane's rows), lanes are combined in fp64."""
        rows, C = y2d.shape
        pad = (-rows) % LANES
        if pad:
            y2d = torch.cat([y2d, torch.zeros((pad, C), dtype=y2d.dtype)])
        t = y2d.view(-1, LANES, C)                      # row r -> lane r % LANES
        s = torch.cumsum(t, 0)[-1].double().sum(0)
        q = torch.cumsum(t * t, 0)[-1].double().sum(0)
        return s, q

    def forward(self, x):
        if not self.training:
            return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, False, 0.0, self.eps)
        B, C = x.shape[0], x.shape[1]
        y2d = x.detach().transpose(0, 1).reshape(C, -1).t().contiguous()      # rows x C like the HIP layout
        n = y2d.shape[0]
        with torch.no_grad():
            if mode == "bn_shift":
                c = self.running_mean.clone()
                s, q = lane_sums(y2d - c)
                mean = c.double() + s / n
                var = (q / n - (s / n) ** 2).clamp_min(0.0)
            else:
                s, q = lane_sums(y2d)
                mean = s / n
                var = (q / n - mean * mean).clamp_min(0.0)
            self.running_mean.mul_(1 - self.momentum).add_(self.momentum * mean.float())
            self.running_var.mul_(1 - self.momentum).add_(self.momentum * (var * n / max(n - 1, 1)).float())
            self.num_batches_tracked += 1
            invstd = (1.0 / torch.sqrt(var + self.eps)).float()
            meanf = mean.float()
        shape = (1, C, 1, 1)
        # the value uses the emulated (mean, invstd); the gradient is that of the true batch normalisation
        xm = x.mean(dim=(0, 2, 3), keepdim=True)
        xv = x.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        exact = (x - xm) / torch.sqrt(xv + self.eps)
        emul = (x.detach() - meanf.view(shape)) * invstd.view(shape)
        xhat = exact + (emul - exact.detach())
        return xhat * self.weight.view(shape) + self.bias.view(shape)

    torch.nn.BatchNorm2d.forward = forward


def run_seeds(variant, first, last, threads):
    import torch
    torch.set_num_threads(threads)
    sys.path.insert(0, HERE)
    import make_golden as G  # noqa: F401  (imports the reference with its placeholders + compiled KNN)
    from randlanet import AugmentationSettings, Model, RandLANetSettings, TrainingSettings
    import randlanet.utils.trainer as T

    parts = set(variant.split("+"))
    flush = "adam_eps0" in parts

    class DenoisedAdam(torch.optim.Adam):
        def step(self, closure=None):
            for group in self.param_groups:
                for p in group["params"]:
                    if p.grad is None:
                        continue
                    if getattr(p, "_true_gradient_is_zero", False):
                        p.grad.zero_()
                    elif flush:
                        g = p.grad
                        g[g.abs() < 1e-7 * g.abs().max()] = 0.0
            return super().step(closure)

    T.torch.optim.Adam = DenoisedAdam
    for m in ("bn_sumsq", "bn_shift"):
        if m in parts:
            _patch_batchnorm(m)
    z = np.load(os.path.join(HERE, "train_run.npz"))
    clouds = [(xyz, np.zeros((xyz.shape[0], 0), np.float32), lab.astype(np.int64)) for xyz, lab in zip(z["clouds"], z["labels"])]
    train, val = clouds[:8], clouds[8:]
    hists = []
    for seed in range(first, last):
        torch.manual_seed(seed)
        np.random.seed(seed)
        s = RandLANetSettings(n_classes=3, n_points=1024, n_neighbors=16, layer_sizes=[8, 16, 32, 32], knn="approximate")
        model = Model(s, use_gpu=False)
        model.module.fc_end[2].p = 0.0
        for name, p in model.module.named_parameters():
            if name.endswith("conv.bias") and not name.startswith("fc_end.3"):
                p._true_gradient_is_zero = True
            if "fcstart" in parts and name == "fc_start.bias":
                p._true_gradient_is_zero = True
        hist = []
        ts = TrainingSettings(epochs=6, batch_size=4, learning_rate=1e-2, early_stopping=False)
        model.train(train, val, ts, AugmentationSettings(), None, ["bg", "a", "b"],
                    callbacks=[lambda e, m: hist.append([m["loss"], m["mIoU"], m["val_loss"], m["val_mIoU"]])])
        hists.append(hist)
        print(f"[{variant}] seed {seed}: val_mIoU {np.round(np.array(hist)[:, 3], 4).tolist()}", flush=True)
    return np.array(hists, dtype=np.float64)


def stat(h):
    v = h[:, :, 3]
    return {"final": v[:, -1], "best": v.max(1), "last3": v[:, -3:].mean(1)}


def report():
    files = sorted(f for f in os.listdir(OUT) if f.endswith(".npz"))
    sets = {f[:-4]: np.load(os.path.join(OUT, f)) for f in files}
    stored = np.load(os.path.join(HERE, "train_seeds_denoised.npz"))
    sets["stored(train_seeds_denoised)"] = stored
    base = sets.get("base")
    if base is None:
        print("no base.npz yet")
        return
    bs = {int(s): i for i, s in enumerate(base["seeds"])}
    print(f"{'variant':32s} {'seeds':>5s}  " + "  ".join(f"{k:>26s}" for k in ("final", "best", "last3", "train loss e6", "val loss e6")))
    for name, z in sets.items():
        common = [int(s) for s in z["seeds"] if int(s) in bs]
        if not common:
            continue
        hi = z["histories"][[list(z["seeds"]).index(s) for s in common]]
        hb = base["histories"][[bs[s] for s in common]]
        a, b = stat(hi), stat(hb)
        cols = []
        for key in ("final", "best", "last3"):
            d = a[key] - b[key]
            se = d.std(ddof=1) / np.sqrt(len(d)) if len(d) > 1 else float("nan")
            cols.append(f"{a[key].mean():.4f} {d.mean():+.4f}+-{se:.4f} ({d.mean() / se:+.1f}s)")
        for col in (0, 2):
            d = hi[:, -1, col] - hb[:, -1, col]
            se = d.std(ddof=1) / np.sqrt(len(d)) if len(d) > 1 else float("nan")
            cols.append(f"{d.mean():+.4f}+-{se:.4f}")
        print(f"{name:32s} {len(common):5d}  " + "  ".join(f"{c:>26s}" for c in cols))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="base")
    ap.add_argument("--seeds", default="0:64")
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--worker", default=None)
    ap.add_argument("--tag", default=None, help="file name under bisect/ (default: the variant)")
    ap.add_argument("--report", action="store_true")
    ap.add_argument("--export", action="store_true",
                    help="write tests/golden/train_seeds_denoised_256.npz: the de-noised reference's histories over 256 seeds, "
                         "two draws (bisect/base.npz: 1 thread per process, bisect/base_t2.npz: 2 threads)")
    ap.add_argument("--fixture", default=None,
                    help="write the merged histories (float32) as tests/golden/<name> instead of bisect/<tag>.npz; "
                         "train_seeds_denoised_fc4096.npz = --variant fcstart --seeds 0:4096 --fixture train_seeds_denoised_fc4096.npz")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    if a.report:
        return report()
    if a.export:
        d1, d2 = np.load(os.path.join(OUT, "base.npz")), np.load(os.path.join(OUT, "base_t2.npz"))
        for z in (d1, d2):
            assert np.array_equal(z["seeds"], np.arange(len(d1["seeds"])))
        np.savez_compressed(os.path.join(HERE, "train_seeds_denoised_256.npz"), seeds=d1["seeds"],
                            histories=np.stack([d1["histories"], d2["histories"]]))
        return
    first, last = (int(v) for v in a.seeds.split(":"))
    if a.worker is not None:
        h = run_seeds(a.variant, first, last, a.threads)
        np.savez_compressed(a.worker, seeds=np.arange(first, last), histories=h)
        return
    # parent: split the seed range over worker processes (each imports the reference on its own)
    bounds = np.linspace(first, last, a.procs + 1).astype(int)
    procs, parts = [], []
    for i in range(a.procs):
        if bounds[i] == bounds[i + 1]:
            continue
        part = os.path.join("/tmp", f"bisect_{a.tag or a.variant}_{bounds[i]}_{bounds[i + 1]}.npz")
        parts.append(part)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--variant", a.variant, "--seeds",
                                       f"{bounds[i]}:{bounds[i + 1]}", "--threads", str(a.threads), "--worker", part]))
    rc = [p.wait() for p in procs]
    assert not any(rc), rc
    zs = [np.load(p) for p in parts]
    seeds, hist = np.concatenate([z["seeds"] for z in zs]), np.concatenate([z["histories"] for z in zs])
    for p in parts:
        os.remove(p)
    if a.fixture:
        np.savez_compressed(os.path.join(HERE, a.fixture), seeds=seeds, histories=hist.astype(np.float32))
        return
    np.savez_compressed(os.path.join(OUT, f"{a.tag or a.variant}.npz"), seeds=seeds, histories=hist)
    report()


if __name__ == "__main__":
    main()
