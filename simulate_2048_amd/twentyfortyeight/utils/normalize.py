"""Logarithmic reward normalisation.  Mirror of ``twentyfortyeight/utils/normalize.py`` (a scalar
host-side formula in the reference too; nothing to offload for one float)."""

from math import log2


def normalize_reward(reward: float, max_tile: int = 2 ** (4**2)) -> float:
    """[ref normalize.py:6-33] log2(reward) / log2(max_tile); 0 stays 0."""
    if reward == 0:
        return 0.0
    return log2(reward) / log2(max_tile)
