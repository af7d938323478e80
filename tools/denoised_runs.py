#!/usr/bin/env python3
"""The de-noised G6 training run (tests/test_model_gpu.py::test_denoised_training_matches_denoised_reference) on the HIP path
over a range of seeds, histories to an .npz - for ablations of the HIP path's own switches (environment) against itself and
for seed ranges beyond the committed fixture:
    [RL_...=1] python tools/denoised_runs.py --seeds 0:1024 --out gpurun_out/den_hip_<tag>.npz [--fc-start]"""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))


class _Patch:
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:256")
    ap.add_argument("--out", required=True)
    ap.add_argument("--fc-start", action="store_true", help="freeze fc_start.bias as well")
    ap.add_argument("--plain", action="store_true", help="no gradient is frozen (the plain run)")
    a = ap.parse_args()
    import logging
    logging.getLogger("trainer").setLevel(logging.WARNING)
    import test_model_gpu as T
    if not a.plain:
        T._freeze_zero_gradient_biases(_Patch(), fc_start=a.fc_start)
    golden = os.path.join(REPO, "tests", "golden")
    first, last = (int(v) for v in a.seeds.split(":"))
    hist = []
    for s in range(first, last):
        hist.append(T._mock_training_run(golden, s)[2])
        if (s - first) % 64 == 63:
            print(f"seed {s} done", flush=True)
    np.savez_compressed(a.out, seeds=np.arange(first, last), histories=np.stack(hist))


if __name__ == "__main__":
    main()
