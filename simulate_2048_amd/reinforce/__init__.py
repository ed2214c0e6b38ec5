"""Mirror of the ``reinforce`` package -- only its environment half (``reinforce.game``) is in scope."""
