"""bench.py's host logic that runs without a GPU: the self-launch path (`python bench.py --gpus N` outside torchrun)
and the sysfs device count its parent uses instead of a HIP call."""
import importlib.util
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _fake_topology(tmp_path, simd_counts):
    for i, simd in enumerate(simd_counts):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {16 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    return str(tmp_path)


def test_visible_gpu_count_reads_sysfs_and_visible_device_lists(bench, tmp_path, monkeypatch):
    topo = _fake_topology(tmp_path, [0, 0, 1024, 1024, 1024, 1024])        # 2 CPU nodes + 4 GPUs
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpu_count(topo) == 4
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpu_count(topo) == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert bench.visible_gpu_count(topo) == 1
    assert bench.visible_gpu_count(str(tmp_path / "missing")) == 0          # no KFD: no GPU, never an exception


def test_launch_ranks_builds_the_torchrun_command_and_propagates_rc(bench, monkeypatch):
    """The positive path of `python bench.py --gpus 4` outside torchrun: one `python -m torch.distributed.run` child with
    one rank per GPU on 127.0.0.1, our own arguments passed through, IPC mode set, and the child's exit code returned."""
    import subprocess
    import types
    calls = []

    def fake_call(cmd, env=None):
        calls.append((cmd, env))
        return 7

    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 8)
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    rc = bench.launch_ranks(types.SimpleNamespace(gpus=4))
    assert rc == 7                                                         # the ranks' exit code is ours
    (cmd, env), = calls
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    port = int(cmd[cmd.index("--master-port") + 1])
    assert 1024 < port < 65536
    script = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and int(env["OMP_NUM_THREADS"]) >= 1
    # fewer devices than ranks: refused with exit code 2, nothing is started
    calls.clear()
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 2)
    assert bench.launch_ranks(types.SimpleNamespace(gpus=4)) == 2 and not calls


def test_roofline_helpers(bench):
    # config A forward fused minimum 103 MB at e = 2, and the fp32 figure the line's whole_step uses
    a2 = bench.fused_min_bytes_per_cloud(40960, 16, [16, 64, 128, 256], 2, 2)
    a4 = bench.fused_min_bytes_per_cloud(40960, 16, [16, 64, 128, 256], 2, 4)
    assert abs(a2 / 1e6 - 103) < 2 and a4 > a2
    ws = bench.whole_step_roofline(bench.CFG, 1000.0)
    assert ws["fused_min_bytes_per_cloud"] == 3 * a4 and abs(ws["frac"] - 1000.0 * 3 * a4 / 8e12) < 1e-4
    assert bench.whole_step_roofline(bench.CFG_S, 100.0)["fused_min_bytes_per_cloud"] > ws["fused_min_bytes_per_cloud"]
