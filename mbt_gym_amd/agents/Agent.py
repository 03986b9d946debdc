import abc

import numpy as np


class Agent(metaclass=abc.ABCMeta):
    """A policy: observation matrix (N, D) -> action matrix (N, A) (reference: mbt_gym/agents/Agent.py:6-12)."""

    @abc.abstractmethod
    def get_action(self, state: np.ndarray) -> np.ndarray:
        pass
