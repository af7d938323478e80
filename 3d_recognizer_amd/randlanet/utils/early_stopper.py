"""Early stopping on a monitored metric with a best-weights snapshot (reference
randlanet/utils/early_stopper.py:12-88)."""
import copy
import logging
from typing import Dict, Optional

import numpy as np

logger = logging.getLogger("early stopper")


class EarlyStopper:
    def __init__(self, patience: int, metric: str, mode: str = "max"):
        assert mode in ("max", "min"), "mode should be max or min!"
        self._patience, self._metric, self._mode = patience, metric, mode
        self.reset()

    def reset(self) -> None:
        self._count = 0
        self._best_model_weights = None
        self._reference = -1 if self._mode == "max" else np.inf

    def check(self, metrics: Dict[str, float], model) -> bool:
        """True while training should go on; snapshots the weights on every non-worse value."""
        if self._metric not in metrics:
            logger.warning(f"Metric {self._metric} not known!")
            return True
        value = metrics[self._metric]
        better = value >= self._reference if self._mode == "max" else value <= self._reference
        if better:
            self._count, self._reference = 0, value
            self._best_model_weights = copy.deepcopy(model.state_dict())
        else:
            self._count += 1
            logger.info(f"No improvement in metric {self._metric} ({self._reference:.3f}) detected for "
                        f"{self._count}/{self._patience} epochs.")
        go_on = self._count < self._patience
        if not go_on:
            logger.info(f"Stopping training as no improvement in {self._metric} was detected for "
                        f"{self._patience} consecutive test runs.")
        return go_on

    def load_best_model_weights(self, model) -> Optional[object]:
        if self._best_model_weights is None:
            return None
        model.load_state_dict(self._best_model_weights)
        logger.info(f"Returning model with {self._metric}: {self._reference:.3f}")
        return model
