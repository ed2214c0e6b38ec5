"""2048 game environment class.  Mirror of ``twentyfortyeight/envs/twentyfortyeight.py``."""

from __future__ import annotations

from numpy import int64, ndarray, zeros

from ..core.gameboard import fill_cells, is_done, next_state
from ..utils.binary import encode_flatten
from ..utils.normalize import normalize_reward


class TwentyFortyEight:
    """
    2048 game environment with the reference's interface [ref twentyfortyeight.py:10-140]:
    ``reset(seed) -> observation``, ``step(action) -> (observation, reward, done)``, properties
    ``observation`` / ``reward`` / ``is_finished``, ``ACTIONS`` and the ``_current_state`` int64
    board that callers may read or overwrite.  Every transition runs on the GPU.
    """

    # ##: All Actions.
    ACTIONS = {'left': 0, 'up': 1, 'right': 2, 'down': 3}

    def __init__(self, size: int = 4, encoded: bool = False, normalize: bool = False):
        if size != 4:
            raise ValueError(f'TwentyFortyEight: the MI355X engine supports size=4 only, got size={size}')
        self.size = size
        self._encoded = encoded
        self._normalize = normalize
        self._current_state: ndarray = zeros((size, size), dtype=int64)
        self._current_reward: float = 0.0
        self.reset()

    @property
    def is_finished(self) -> bool:
        """True when no move is possible [ref :44-54]."""
        return is_done(self._current_state)

    @property
    def observation(self) -> ndarray:
        """Raw board, or the 16 x 31 one-hot log2 encoding when ``encoded`` [ref :56-68]."""
        if self._encoded:
            return encode_flatten(self._current_state, encodage_size=31)
        return self._current_state

    @property
    def reward(self) -> float:
        """Last reward, log-normalised when ``normalize`` [ref :70-82]."""
        if self._normalize:
            return normalize_reward(self._current_reward)
        return self._current_reward

    def reset(self, seed: int | None = None) -> ndarray:
        """Empty board plus two random tiles; returns the observation [ref :84-104]."""
        self._current_state = zeros((self.size, self.size), dtype=int64)
        self._current_state = fill_cells(state=self._current_state, number_tile=2, seed=seed)
        self._current_reward = 0
        return self.observation

    def step(self, action: int) -> tuple[ndarray, float, bool]:
        """Apply `action` (0 left, 1 up, 2 right, 3 down) -> (observation, reward, done) [ref :106-133]."""
        self._current_state, self._current_reward = next_state(state=self._current_state, action=action)
        return self.observation, self.reward, self.is_finished

    def render(self) -> None:
        """Print the board, tab separated [ref :135-140]."""
        for row in self._current_state.tolist():
            print(' \t'.join(map(str, row)))
