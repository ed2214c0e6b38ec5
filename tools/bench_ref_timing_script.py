#!/usr/bin/env python3
"""The reference's own timing script (tests/test_perf_utils.py: slide_and_merge and illegal_actions of single random boards of
sizes 4 / 6 / 8, min-of-repeats average per call) on the API mirrors, GPU box.  Same board generator and protocol, fewer
boards and calls (20 boards x 3 repeats x 300 calls) so that it finishes in seconds.
`--module twentyfortyeight` with the reference on PYTHONPATH times the reference's NumPy functions the same way (build
container only: the reference does not travel to the GPU box)."""
import argparse
import importlib
import json
import sys
import timeit
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
parser = argparse.ArgumentParser()
parser.add_argument('--module', default='simulate_2048_amd.twentyfortyeight')
args = parser.parse_args()
gameboard = importlib.import_module(args.module + '.core.gameboard')
gamemove = importlib.import_module(args.module + '.core.gamemove')
generator = np.random.default_rng(42)


def random_board(size):
    board = np.zeros((size, size), dtype=np.int64)
    num_tiles = generator.integers(1, size * size + 1)
    values = generator.choice([2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048], size=num_tiles)
    board.flat[generator.choice(size * size, size=num_tiles, replace=False)] = values
    return board


def per_call_us(func, board, number=300, repeat=3):
    return min(timeit.Timer(lambda: func(board)).repeat(repeat=repeat, number=number)) / number * 1e6


out = {'module': args.module, 'unit': 'microseconds per call, mean over 20 random boards of min-of-3 x 300 calls'}
for name, func in (('slide_and_merge', gameboard.slide_and_merge), ('illegal_actions', gamemove.illegal_actions)):
    for size in (4, 6, 8):
        boards = [random_board(size) for _ in range(20)]
        func(boards[0])
        out[f'{name} {size}x{size}'] = round(float(np.mean([per_call_us(func, b) for b in boards])), 2)
print(json.dumps(out, indent=1))
