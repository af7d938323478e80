"""CPU-side checks of the C-ABI boundary: the library loads without a GPU, exports every symbol
include/rl_randlanet.h declares, and the ctypes table binds exactly that set."""
import ctypes
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "rl_randlanet.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rl_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from randlanet import _hip
    if not os.path.exists(_hip.library_path()):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "3d_recognizer_amd", "csrc"), "-j4"])
    return ctypes.CDLL(_hip.library_path())


def test_header_symbols_are_exported(lib):
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rl_randlanet.h but not exported"


def test_ctypes_table_matches_header():
    from randlanet import _hip
    assert sorted(_hip.EXPORTS) == _declared()


def test_host_only_entry_points(lib):
    from randlanet import _hip
    L = _hip.lib()
    assert L.rl_version() == _hip.ABI_VERSION == 110
    assert L.rl_row_blocks(1, 128) == 1 and L.rl_row_blocks(128 * 5000, 128) == 1024
    assert L.rl_row_blocks(129, 128) == 2
    assert L.rl_wgrad_slab_floats(1000, 16, 16) > 0
    assert L.rl_loss_work_doubles(10, 2) == 1025 * 11
    # argument validation happens on the host, before any launch
    assert L.rl_knn_f32(None, None, 1, 3, 3, 4, None, None, None, 0, None) == _hip.ERR_FEW_SUPPORT
    assert b"Not enough points" in L.rl_last_error()
    assert L.rl_knn_f32(None, None, 1, 100, 3, 65, None, None, None, 0, None) == _hip.ERR_UNSUPPORTED
    assert L.rl_knn_workspace_bytes(4, 40960, 40960, 16) > 4 * 40960 * 16 and L.rl_knn_workspace_bytes(1, 100, 100, 4) == 0
    assert L.rl_gemm(None, None) == _hip.ERR_ARGS


def test_missing_library_fails_loudly(monkeypatch):
    from randlanet import _hip
    monkeypatch.setattr(_hip, "_LIB", None)
    monkeypatch.setattr(_hip, "_LIB_PATH", "/nonexistent/librandla_hip.so")
    with pytest.raises(_hip.HipKernelError, match="no fallback"):
        _hip.lib()


def test_operands_on_another_device_are_refused_before_launch(monkeypatch):
    """Launches go to the CURRENT device's stream: a tensor on another GPU, or on the host, must raise on the host side
    (a kernel dereferencing another device's pointer is a memory fault on an 8-GPU node)."""
    import torch

    from randlanet import _hip, _ops

    class FakeDeviceTensor:
        is_cuda = True

        def __init__(self, index):
            self._i = index

        def get_device(self):
            return self._i

        def is_contiguous(self):
            return True

    monkeypatch.setattr(_hip, "current_device", lambda: 0)
    _ops._dev_check(FakeDeviceTensor(0), None)
    with pytest.raises(_hip.HipKernelError, match="current device is cuda:0"):
        _ops._dev_check(FakeDeviceTensor(0), FakeDeviceTensor(1))
    with pytest.raises(_hip.HipKernelError, match="device tensors"):
        _ops._dev_check(torch.zeros(3))


def test_ctypes_structures_have_the_headers_layout(tmp_path):
    """Every descriptor struct of include/rl_randlanet.h against its ctypes mirror in randlanet/_hip.py: size and the offset
    of every field, as gcc lays the header out (a field added on one side only shifts everything behind it silently)."""
    from randlanet import _hip
    pairs = {"rl_gemm_desc": _hip.GemmDesc, "rl_wsplit_item": _hip.WsplitItem, "rl_wgrad_desc": _hip.WgradDesc,
             "rl_wgrad_reduce_item": _hip.WgradReduceItem, "rl_bn_bwd_desc": _hip.BnBwdDesc, "rl_knn_task": _hip.KnnTask,
             "rl_pool_desc": _hip.PoolDesc, "rl_resid_bn_bwd_desc": _hip.ResidBnBwdDesc, "rl_csr_task": _hip.CsrTask,
             "rl_bn_finalize_item": _hip.BnFinalizeItem, "rl_head_desc": _hip.HeadDesc,
             "rl_segsum_desc": _hip.SegsumDesc, "rl_rows_desc": _hip.RowsDesc, "rl_cloud_job": _hip.CloudJob}
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    expect = {}
    for cname, ct in pairs.items():
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), text, flags=re.S)
        assert body, cname
        # field names in declaration order: identifiers right before ';', ',' or '['
        fields = []
        for decl in body.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?\s*$", part.strip())
                assert m, (cname, part)
                fields.append(m.group(1))
        assert fields == [f[0] for f in ct._fields_], (cname, fields, [f[0] for f in ct._fields_])
        lines.append(f'printf("{cname} %zu", sizeof({cname}));')
        for f in fields:
            lines.append(f'printf(" %zu", offsetof({cname}, {f}));')
        lines.append('printf("\\n");')
        expect[cname] = [ctypes.sizeof(ct)] + [getattr(ct, f).offset for f in fields]
    lines += ["return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)]).decode().strip().splitlines()
    assert len(out) == len(pairs)
    for line in out:
        name, *nums = line.split()
        assert [int(n) for n in nums] == expect[name], (name, nums, expect[name])
