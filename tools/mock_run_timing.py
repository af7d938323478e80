#!/usr/bin/env python3
"""Where the 0.14 s of one G6 mock training run go (tests/test_model_gpu.py::_mock_training_run): graph capture vs eager launches."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import logging
logging.getLogger("trainer").setLevel(logging.WARNING)
import numpy as np, torch
import test_model_gpu as T
from randlanet.utils import trainer as TR
golden = os.path.join(REPO, "tests", "golden")
def run(n=32):
    T._mock_training_run(golden, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hs = [T._mock_training_run(golden, s)[2] for s in range(n)]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, np.stack(hs)
t_a, h_a = run()
os.environ["RL_EVAL_EAGER"] = "1"
t_b, h_b = run()
orig = TR.Trainer._make_stepper
TR.Trainer._make_stepper = staticmethod(lambda model, B, N, loss, use_graph, state: orig(model, B, N, loss, False, state))
t_c, h_c = run()
print(f"per run: graphs {t_a*1e3:.1f} ms; eval eager {t_b*1e3:.1f} ms; eval + train eager {t_c*1e3:.1f} ms; same bits: {np.array_equal(h_a, h_b)} {np.array_equal(h_a, h_c)}")
