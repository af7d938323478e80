"""K nearest neighbours on the MI355X behind the reference's operator names.

`knn(support, querry, k)` is the drop-in for the compiled `knn_tpk.knn` of the reference
(randlanet/utils/src/bindings.cpp:5-7, knn.cpp:43-61): same argument names (the second one
really is spelled `querry`), same (int64 indices, fp32 squared distances) result, same error
behaviour for too few support points - but it accepts device tensors and runs rl_knn_f32.
`knn_naive` / `knn_approximate` keep the names of randlanet/utils/knn.py:7-117 and return the
exact answer those searches approximate.
"""
from typing import Tuple

import torch

from .. import _hip as H
from .. import _ops as ops


def _device_of(*ts) -> torch.device:
    for t in ts:
        if t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise H.HipKernelError("knn needs an MI355X (HIP) device: there is no CPU path in this build")
    return torch.device("cuda")


def knn_exact(xyz: torch.Tensor, xyz_query: torch.Tensor, n_neighbors: int, device=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(B,N',3),(B,N,3) -> neighbours int64 (B,N,K), squared distances fp32 (B,N,K), ascending,
    ties by lowest index; results live on the device.  `device` = the owning module's device: a module placed on the
    CPU searches on the host."""
    on_host = (device is not None and torch.device(device).type != "cuda") or \
        (not xyz.is_cuda and not xyz_query.is_cuda and not torch.cuda.is_available())
    if on_host:
        # a box without a HIP device: the library's host twin (rl_knn_f32_cpu), as the reference's CPU-only knn_tpk.knn
        from .._cpu import knn_host
        try:
            return knn_host(xyz, xyz_query, int(n_neighbors))
        except H.HipKernelError as e:
            if "Not enough points" in str(e):
                raise RuntimeError(f"Not enough points in support to find {n_neighbors} neighboors") from e
            raise
    dev = _device_of(xyz, xyz_query)
    s = xyz.to(dev, torch.float32).contiguous()
    q = xyz_query.to(dev, torch.float32).contiguous()
    try:
        with torch.cuda.device(dev):
            return ops.knn_f32(s, q, int(n_neighbors))
    except H.HipKernelError as e:
        if "Not enough points" in str(e):
            # the reference raises c10::Error -> RuntimeError with this text (knn.cpp:15-17)
            raise RuntimeError(f"Not enough points in support to find {n_neighbors} neighboors") from e
        raise


def knn(support: torch.Tensor, querry: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Drop-in for knn_tpk.knn.  Contiguity is required as in the reference (knn.cpp:45-46);
    the reference's CPU-only check (knn.cpp:47-48) is lifted: tensors may live on the device,
    host tensors are copied over, and the result comes back on the input's device."""
    if not support.is_contiguous():
        raise RuntimeError("support must be a contiguous tensor")
    if not querry.is_contiguous():
        raise RuntimeError("query must be a contiguous tensor")
    idx, d2 = knn_exact(support, querry, k)
    return idx.to(querry.device), d2.to(querry.device)


def knn_naive(xyz, xyz_query, n_neighbors, partition_size: int = 4000, n_parts_max: int = 15):
    """Reference knn.py:7-55 (distance-matrix top-k); here exact, no slabs needed."""
    return knn_exact(xyz, xyz_query, n_neighbors)


def knn_approximate(xyz, xyz_query, n_neighbors):
    """Reference knn.py:58-117 (FAISS IVF on the CPU); here the exact search it approximates."""
    return knn_exact(xyz, xyz_query, n_neighbors)
