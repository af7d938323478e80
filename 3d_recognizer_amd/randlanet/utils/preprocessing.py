"""Point sampling used by the data loader and by Model.predict (reference
randlanet/utils/preprocessing.py:6-62).  Host-side numpy glue: the values and the way the global
numpy RNG is consumed match the reference, so "consistent" (seed 0) sampling picks the same
points as the reference does."""
from contextlib import contextmanager
from typing import Optional

import numpy as np


@contextmanager
def _fixed_seed(active: bool, seed: int = 0):
    """Inside: numpy's global RNG restarts from `seed`; afterwards its previous state is back."""
    if not active:
        yield
        return
    state = np.random.get_state()
    np.random.seed(seed)
    try:
        yield
    finally:
        np.random.set_state(state)


def random_choice(a: int, size: int, replace: bool = True, p: Optional[np.ndarray] = None,
                  consistent: bool = False) -> np.ndarray:
    """np.random.choice, optionally from a freshly seeded (0) generator (preprocessing.py:6-32)."""
    with _fixed_seed(consistent):
        return np.random.choice(a, size, replace, p)


def sample_points(n_points: int, n_sample_points: int, consistent: bool = False) -> np.ndarray:
    """Indices of a random sub-sample; when more points are requested than exist, every point is
    taken once and the remainder is drawn with replacement (preprocessing.py:35-62)."""
    take = min(n_sample_points, n_points)
    picked = random_choice(n_points, take, replace=False, consistent=consistent)
    missing = n_sample_points - n_points
    if missing > 0:
        extra = random_choice(n_points, missing, replace=True, consistent=consistent)
        picked = np.r_[picked, extra]
    return picked
