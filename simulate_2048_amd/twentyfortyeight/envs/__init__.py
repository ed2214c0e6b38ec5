"""2048 game environment for reinforcement learning agents (mirror of ``twentyfortyeight/envs``)."""
