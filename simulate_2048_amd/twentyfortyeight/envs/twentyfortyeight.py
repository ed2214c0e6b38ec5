"""
Single-board environment object with the public surface of the reference's
``twentyfortyeight/envs/twentyfortyeight.py`` (class ``TwentyFortyEight``), backed by the HIP kernels.

Surface kept (reference lines in brackets): constructor flags ``size / encoded / normalize`` [:21] (size 4: the packed
kernels; 2 ... 8: the general ``env2048_nxn_*`` kernels),
``ACTIONS`` [:19], ``reset(seed)`` [:84-104], ``step(action)`` [:106-133], the properties
``observation`` [:56-68], ``reward`` [:70-82], ``is_finished`` [:44-54], ``render()`` [:135-140] and the
two attributes callers are known to poke, ``_current_state`` (int64 4x4 values) and
``_current_reward`` (tests/test_environment.py:106-107,256,269,315,335 of the reference).
"""

from __future__ import annotations

import numpy as np

from ..core import gameboard
from ..utils import binary, normalize

_ONE_HOT_WIDTH = 31  # the reference's encodage_size for the flattened observation


class TwentyFortyEight:
    ACTIONS = {'left': 0, 'up': 1, 'right': 2, 'down': 3}

    def __init__(self, size: int = 4, encoded: bool = False, normalize: bool = False):
        if not isinstance(size, (int, np.integer)) or not 2 <= size <= 8:
            raise ValueError(f'TwentyFortyEight: the MI355X engine supports size = 2 ... 8, got size={size}')
        self.size = int(size)
        self._encoded = bool(encoded)
        self._normalize = bool(normalize)
        self._current_state = np.zeros((self.size, self.size), dtype=np.int64)
        self._current_reward = 0.0
        self.reset()

    # -- what the agent sees ---------------------------------------------------------------------
    @property
    def observation(self) -> np.ndarray:
        """The board itself, or its 16 x 31 one-hot log2 code when built with ``encoded=True``."""
        board = self._current_state
        return binary.encode_flatten(board, encodage_size=_ONE_HOT_WIDTH) if self._encoded else board

    @property
    def reward(self) -> float:
        """Score of the last move; log-scaled into [0, 1] when built with ``normalize=True``."""
        raw = self._current_reward
        return normalize.normalize_reward(raw) if self._normalize else raw

    @property
    def is_finished(self) -> bool:
        """No move changes the board any more (evaluated on the current ``_current_state``)."""
        return gameboard.is_done(self._current_state)

    # -- transitions ------------------------------------------------------------------------------
    def reset(self, seed: int | None = None) -> np.ndarray:
        """Start over: an empty board that receives two tiles. Returns the new observation."""
        fresh = np.zeros((self.size, self.size), dtype=np.int64)
        self._current_state = gameboard.fill_cells(state=fresh, number_tile=2, seed=seed)
        self._current_reward = 0
        return self.observation

    def step(self, action: int) -> tuple[np.ndarray, float, bool]:
        """Play ``action`` (0 left, 1 up, 2 right, 3 down): ``(observation, reward, finished)``."""
        # next_state + is_done of the reference in ONE launch: the step kernel reports the terminal test
        # of the board it wrote.  An invalid move keeps the same state object and earns the int 0.
        # With ``encoded=True`` the one-hot observation of the new board comes from a second launch right behind the step
        # (it reads the exponents the step has just written) and both are waited for once.
        width = _ONE_HOT_WIDTH if self._encoded else 0
        board, gained, finished, *onehot = gameboard._transition(self._current_state, action, None, width)
        if np.array_equal(board, self._current_state):
            board, gained = self._current_state, 0
        else:
            gained = 0.0 + float(gained)
        self._current_state, self._current_reward = board, gained
        observation = onehot[0].astype(np.int64).ravel() if onehot else self.observation
        return observation, self.reward, finished

    def render(self) -> None:
        """Print the four rows, cells separated by ' \\t'."""
        for row in self._current_state.tolist():
            print(' \t'.join(str(cell) for cell in row))
