"""Annotation glue of the reference's dataset loader (reference dataset.py:8-18).

`broaden_annotation(point_cloud, annotation, radius)` marks every point that lies closer than `radius` to ANY annotated
point.  The reference does it with one Python iteration (a full-cloud norm) per annotated point and `np.bool`, which
numpy removed in 1.24; this is the same predicate evaluated in blocks - same dtype, same operation order
(sqrt(((dx*dx)+(dy*dy))+(dz*dz)) < radius, what np.linalg.norm(..., axis=1) computes), so the mask is bit-identical.
A maintainer replaces the body of the reference function with `from randlanet.utils.annotation import broaden_annotation`.
"""
import numpy as np

_BLOCK_ELEMENTS = 1 << 22     # annotated points x cloud points per block (~50 MB of float32 differences)


def broaden_annotation(point_cloud: np.ndarray, annotation: np.ndarray, radius: float = 0.01) -> np.ndarray:
    cloud = np.asarray(point_cloud)
    if not np.issubdtype(cloud.dtype, np.inexact):
        cloud = cloud.astype(float)                       # np.linalg.norm promotes integer input the same way
    picked = cloud[np.asarray(annotation).astype(bool)]
    if picked.shape[0] == 0:
        return np.logical_or.reduce([]).astype(np.uint8)  # what the reference returns for an empty annotation
    n = cloud.shape[0]
    out = np.zeros(n, dtype=bool)
    step = max(1, _BLOCK_ELEMENTS // max(n, 1))
    for a0 in range(0, picked.shape[0], step):
        diff = picked[a0:a0 + step, None, :] - cloud[None, :, :]
        dist = np.sqrt(np.add.reduce(diff * diff, axis=-1))
        out |= (dist < radius).any(axis=0)
    return out.astype(np.uint8)
