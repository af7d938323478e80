#!/usr/bin/env python3
"""Paired comparison of de-noised training runs: reference histories (tests/golden/train_seeds_denoised_fc*.npz, made by
tests/golden/bisect_probe.py --variant fcstart --fixture ...) against HIP histories (tools/denoised_runs.py --fc-start), by seed.
    python tools/denoised_compare.py --ref tests/golden/train_seeds_denoised_fc4096.npz [more ...] --hip profiles/x.npz [more ...] [--mode fp32]
Prints mean reference / HIP validation mIoU, the paired difference, its standard error and the same per block of 4096 seeds,
for the final / best / last-three epochs (the three readings tests/test_model_gpu.py asserts)."""
import argparse

import numpy as np


def load(paths, key="bf16x3"):
    seeds, hist = [], []
    for p in paths:
        z = np.load(p)
        h = z["histories"] if "histories" in z else z[key]      # (profiles/r0x_denoised_hip_*.npz: one array per arithmetic mode)
        if h.ndim == 4:          # (draws, S, epochs, 4): first draw
            h = h[0]
        seeds.append(z["seeds"])
        hist.append(h.astype(np.float64))
    seeds, hist = np.concatenate(seeds), np.concatenate(hist)
    order = np.argsort(seeds, kind="stable")
    assert len(np.unique(seeds)) == len(seeds), "a seed appears twice"
    return seeds[order], hist[order]


def views(h):
    v = h[:, :, 3]               # validation mIoU per epoch
    return {"final": v[:, -1], "best": v.max(1), "last3": v[:, -3:].mean(1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", nargs="+", required=True)
    ap.add_argument("--hip", nargs="+", required=True)
    ap.add_argument("--block", type=int, default=4096)
    ap.add_argument("--mode", default="bf16x3", help="which array of a HIP file that holds one per arithmetic mode (bf16x3 | fp32)")
    a = ap.parse_args()
    rs, rh = load(a.ref)
    hs, hh = load(a.hip, a.mode)
    common = np.intersect1d(rs, hs)
    rh, hh = rh[np.searchsorted(rs, common)], hh[np.searchsorted(hs, common)]
    print(f"{len(common)} paired seeds ({common[0]} .. {common[-1]})")
    R, Hh = views(rh), views(hh)
    for key in ("final", "best", "last3"):
        d = Hh[key] - R[key]
        se = d.std(ddof=1) / np.sqrt(len(d))
        print(f"val mIoU [{key}]: reference {R[key].mean():.4f}, hip {Hh[key].mean():.4f} +- {Hh[key].std(ddof=1):.4f}; "
              f"paired difference {100 * d.mean():+.3f} pt (SE {100 * se:.3f} pt, {d.mean() / se:+.2f} sigma)")
        blocks = [d[i:i + a.block] for i in range(0, len(d), a.block) if len(d[i:i + a.block]) == a.block]
        if len(blocks) > 1:
            print("    per block of %d: " % a.block + ", ".join(f"{100 * b.mean():+.3f}" for b in blocks) + " pt")
    for name, col in (("validation loss", 2), ("training loss", 0)):
        d = hh[:, -1, col] - rh[:, -1, col]
        print(f"{name} (last epoch): paired difference {d.mean():+.5f} (SE {d.std(ddof=1) / np.sqrt(len(d)):.5f})")


if __name__ == "__main__":
    main()
