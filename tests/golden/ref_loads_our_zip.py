#!/usr/bin/env python3
"""Model-zip compatibility, the OTHER direction: a zip written by THIS package's Model.save is loaded
by the REFERENCE's Model.load (reference randlanet/model.py:77-105) and predicts the same confidences.

Build-container only (needs /root/reference and oracle/_ref/knn_tpk.so).  Both packages are called `randlanet`, so
each side runs in its own interpreter:
    python tests/golden/ref_loads_our_zip.py            # drives both steps, prints the comparison, exit code 0 = ok
    ... write <zip> <npz>    (this package, CPU device: saves a model with formula weights + its own predictions)
    ... check <zip> <npz>    (the reference: loads the zip, compares settings, state_dict and predictions)
"""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("REF_ROOT", "/root/reference")
SETTINGS = dict(n_classes=3, n_points=800, n_neighbors=8, layer_sizes=[8, 16, 32, 32], knn="approximate", upsampling="idw")


def write(zip_path, npz_path):
    sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
    sys.path.insert(0, REPO)
    from pathlib import Path

    import numpy as np
    from oracle.init_formula import formula_state_dict          # weights by formula: the checker rebuilds them
    from randlanet import Model, RandLANetSettings
    model = Model(RandLANetSettings(**SETTINGS), use_gpu=False)
    sd = model.module.state_dict()
    model.module.load_state_dict(formula_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], seed=9))
    model.module.eval()
    model.save(Path(zip_path))
    cloud = np.random.RandomState(4).uniform(0, 1, (3000, 3)).astype(np.float32)
    np.random.seed(77)
    conf = model.predict(cloud)
    np.savez(npz_path, cloud=cloud, conf=conf)


def check(zip_path, npz_path):
    import types
    from pathlib import Path

    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(REPO, "oracle", "_ref"))
    sys.path.insert(0, REPO)
    import knn_tpk
    sys.modules.setdefault("faiss", types.ModuleType("faiss"))     # inert placeholders, as in make_golden.py
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    sys.path.insert(0, REF)
    from randlanet import Model
    from randlanet.utils import modules as M
    assert M.__file__.startswith(REF), M.__file__
    M.knn_approximate = lambda s, q, k: knn_tpk.knn(s.contiguous().float().cpu(), q.contiguous().float().cpu(), k)
    from oracle.init_formula import formula_state_dict
    model = Model.load(Path(zip_path), use_gpu=False)               # the reference's loader on our file
    s = model.settings
    for key, val in SETTINGS.items():
        assert getattr(s, key) == val, (key, getattr(s, key), val)
    sd = model.module.state_dict()
    want = formula_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], seed=9)
    assert list(sd.keys()) == list(want.keys())
    for k in sd:
        assert torch.equal(sd[k].cpu(), want[k]), k
    z = np.load(npz_path)
    np.random.seed(77)
    conf = np.asarray(model.predict(z["cloud"]))
    err = np.abs(conf - z["conf"]).max()
    print(f"reference loaded our zip: {len(sd)} state_dict entries identical, settings identical, "
          f"max |confidence difference| on a 3000-point cloud = {err:.2e}")
    assert err < 1e-4, err


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "write":
        write(sys.argv[2], sys.argv[3])
    elif len(sys.argv) == 4 and sys.argv[1] == "check":
        check(sys.argv[2], sys.argv[3])
    else:
        with tempfile.TemporaryDirectory() as d:
            z, n = os.path.join(d, "ours_model"), os.path.join(d, "ours.npz")
            env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
            subprocess.check_call([sys.executable, __file__, "write", z, n], env=env)
            subprocess.check_call([sys.executable, __file__, "check", z, n], env=env)
