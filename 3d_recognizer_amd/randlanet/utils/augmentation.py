"""Training-time point-cloud augmentation (reference randlanet/utils/augmentation.py:7-167):
jitter -> scale -> rotate -> shift, per sample, on the host.  The random draws happen in the
reference's order (randn(N,3); uniform; 3 x randn; uniform(3)) so a seeded run augments
identically."""
from dataclasses import dataclass
from typing import Tuple

import numpy as np


@dataclass
class AugmentationSettings:
    #: Variance of the per-point jitter (scaled by the mean cloud radius)
    jitter_variance: float = 0.01
    #: Clip value of the per-point jitter
    jitter_limit: float = 0.05
    #: Scale is drawn from [1 - scale_limit, 1 + scale_limit]
    scale_limit: float = 0.2
    #: Largest shift, in mean cloud radii
    shift_limit: float = 0.1
    #: Variances of the random rotation angles around x, y, z (rad)
    rotation_angle_variances: Tuple[float, float, float] = (0.06, 0.06, 0.06)
    #: Clip values of those angles (rad)
    rotation_angle_limits: Tuple[float, float, float] = (0.18, 0.18, 0.18)


def _centre(xyz: np.ndarray) -> np.ndarray:
    return np.mean(xyz, axis=0, keepdims=True)


def get_mean_radius(xyz: np.ndarray) -> float:
    """Mean distance to the centroid (augmentation.py:24-33)."""
    return float(np.mean(np.linalg.norm(xyz - _centre(xyz), axis=1)))


def jitter_point_cloud(xyz: np.ndarray, variance: float = 0.01, limit: float = 0.05) -> np.ndarray:
    noise = get_mean_radius(xyz) * variance * np.random.randn(*xyz.shape)
    return np.clip(noise, -limit, limit) + xyz


def random_scale_point_cloud(xyz: np.ndarray, scale_limit: float = 0.2) -> np.ndarray:
    factor = np.random.uniform(1 - scale_limit, 1 + scale_limit)
    c = _centre(xyz)
    return (xyz - c) * factor + c


def _rotation(ax: float, ay: float, az: float) -> np.ndarray:
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def random_rotate_point_cloud(xyz: np.ndarray, angle_variances=(0.06, 0.06, 0.06),
                              angle_limits=(0.18, 0.18, 0.18)) -> np.ndarray:
    assert len(angle_variances) == 3, "angle_sigmas should have length 3"
    assert len(angle_limits) == 3, "angle_clips should have length 3"
    angles = [float(np.clip(s * np.random.randn(), -lim, lim)) for s, lim in zip(angle_variances, angle_limits)]
    c = _centre(xyz)
    return (xyz - c) @ _rotation(*angles).T + c


def random_shift_point_cloud(xyz: np.ndarray, shift_limit: float = 0.1) -> np.ndarray:
    return xyz + get_mean_radius(xyz) * np.random.uniform(-shift_limit, shift_limit, 3)


def perturbate_point_cloud(xyz: np.ndarray, settings: AugmentationSettings) -> np.ndarray:
    out = jitter_point_cloud(xyz, settings.jitter_variance, settings.jitter_limit)
    out = random_scale_point_cloud(out, settings.scale_limit)
    out = random_rotate_point_cloud(out, settings.rotation_angle_variances, settings.rotation_angle_limits)
    return random_shift_point_cloud(out, settings.shift_limit)
