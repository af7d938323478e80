#!/usr/bin/env python3
"""Print the rl_gemm / rl_wgrad descriptors of one eager training step of config A with >= <thr> rows (debug aid)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import numpy as np, torch
import bench
from randlanet import _hip as H
from randlanet._train import TrainStep
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
lib = H.lib()
class Spy:
    def __getattr__(self, k):
        f = getattr(lib, k)
        if k == "rl_gemm":
            def g(dref, st):
                d = dref._obj
                M = d.B * d.n * (d.nbr_k if d.a_mode == 1 else 1)
                if M >= thr and d.K >= 64:
                    print(f"gemm M={M} K={d.K} N={d.N} lda={d.lda} a_bstride={d.a_bstride} n={d.n} lazy={bool(d.in_scale)} act={d.in_act} "
                          f"w_ks={d.w_ks} w_ns={d.w_ns} ldy={d.ldy} y_bstride={d.y_bstride} acc={d.accumulate} stats={bool(d.stats)} bias={bool(d.bias)}")
                return f(dref, st)
            return g
        if k == "rl_wgrad":
            def g(dref, st):
                d = dref._obj
                M = d.B * d.n * (d.nbr_k if d.a_mode == 1 else 1)
                if M >= thr and d.K >= 64:
                    print(f"wgrad M={M} K={d.K} N={d.N} lda={d.lda} a_bstride={d.a_bstride} n={d.n} lazy={bool(d.in_scale)} lddy={d.lddy} dy_bstride={d.dy_bstride}")
                return f(dref, st)
            return g
        return f
spy = Spy()
H.lib = lambda: spy
dev = torch.device("cuda")
B, N, C = bench.CFG["per_gpu_batch"], bench.CFG["n_points"], bench.CFG["n_classes"]
model = bench.build_model(dev, seed=0); model.train()
st = TrainStep(model, B, N, loss="dice", lr=1e-2, use_graph=False, world_size=1)
xyz, labels = bench.synthetic_batch(B, N, C, 1234)
st.set_batch(torch.from_numpy(xyz).to(dev), torch.from_numpy(labels).to(dev))
st.capture()
print("---- one step ----")
st.step(np.random.permutation(N)); torch.cuda.synchronize()
