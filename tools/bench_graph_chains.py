#!/usr/bin/env python3
"""Short sequences (K = 20 ... 64 steps of 2^20 boards between two synchronisations): K launches enqueued by one
env2048_step_sequence call vs the same launches captured once into a HIP graph and replayed -- as one chain, and as two
half-batch chains on two branches of the graph.  Timing only (the captured launches carry a fixed step index; the product form
uses the device step counter).  Run on the GPU box:

    python3 tools/bench_graph_chains.py [out.json]
"""
import ctypes
import json
import statistics
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import BatchedEnv2048, capi  # noqa: E402

N = 1 << 20
HALF = N // 2


def main():
    lib = capi.library()
    dev = torch.device('cuda', 0)
    env = BatchedEnv2048(N, 7, device=dev, auto_reset=True)
    env.reset()
    env.rollout(64)  # mid-game boards
    torch.cuda.synchronize()
    rows = {}
    for K in (20, 32, 64):
        acts = torch.randint(0, 4, (K, N), dtype=torch.uint8, device=dev)
        base = acts.data_ptr()
        table = (ctypes.c_void_p * K)(*[base + k * N for k in range(K)])
        table_b = (ctypes.c_void_p * K)(*[base + k * N + HALF for k in range(K)])
        b, r, d = env.boards.data_ptr(), env.reward.data_ptr(), env.done.data_ptr()
        flags = capi.FLAG_AUTO_RESET

        def eager(extra=0):
            s = torch.cuda.current_stream().cuda_stream
            lib.step_sequence(0, s, N, 0, b, table, r, d, None, None, 0, 7, 100, K, flags | extra)

        def one_chain_body():
            s = torch.cuda.current_stream().cuda_stream
            lib.step_sequence(0, s, N, 0, b, table, r, d, None, None, 0, 7, 100, K, flags | capi.FLAG_SINGLE_CHAIN)

        side = torch.cuda.Stream(dev)

        def two_chain_body():
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                lib.step_sequence(0, side.cuda_stream, HALF, HALF, b + HALF * 16, table_b, r + HALF * 4, d + HALF, None, None, 0, 7, 100, K,
                                  flags | capi.FLAG_SINGLE_CHAIN)
            lib.step_sequence(0, cur.cuda_stream, HALF, 0, b, table, r, d, None, None, 0, 7, 100, K, flags | capi.FLAG_SINGLE_CHAIN)
            cur.wait_stream(side)

        def half_a_body():
            s = torch.cuda.current_stream().cuda_stream
            lib.step_sequence(0, s, HALF, 0, b, table, r, d, None, None, 0, 7, 100, K, flags | capi.FLAG_SINGLE_CHAIN)

        def half_b_body():
            s = torch.cuda.current_stream().cuda_stream
            lib.step_sequence(0, s, HALF, HALF, b + HALF * 16, table_b, r + HALF * 4, d + HALF, None, None, 0, 7, 100, K, flags | capi.FLAG_SINGLE_CHAIN)

        graphs = {}
        for name, body in (('graph_one_chain', one_chain_body), ('graph_two_chains', two_chain_body), ('half_a', half_a_body), ('half_b', half_b_body)):
            g = torch.cuda.CUDAGraph()
            cap = torch.cuda.Stream(dev)
            with torch.cuda.stream(cap):
                body()  # warm
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=cap):
                body()
            graphs[name] = g
        def two_graphs_two_streams():  # ONE thread, two graph launches: each half-batch chain is a graph of its own on its own stream
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                graphs['half_b'].replay()
            graphs['half_a'].replay()
            cur.wait_stream(side)

        forms = {
            'two_graphs_two_streams': two_graphs_two_streams,
            'launches_one_chain': lambda: eager(capi.FLAG_SINGLE_CHAIN),
            'launches_as_planned': lambda: eager(0),
            'graph_one_chain': graphs['graph_one_chain'].replay,
            'graph_two_chains': graphs['graph_two_chains'].replay,
        }
        for rnd in range(2):
            for name, fn in forms.items():
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                walls = []
                for _ in range(200):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    walls.append((time.perf_counter() - t0) * 1e6 / K)
                rows[f'K{K}_{name}_{rnd + 1}'] = {'wall_us_per_step_median': round(statistics.median(walls), 3),
                                                   'min': round(min(walls), 3), 'board_steps_per_s': round(N / statistics.median(walls) * 1e6, -6)}
                print(K, name, rows[f'K{K}_{name}_{rnd + 1}'], flush=True)
    out = {'what': 'K steps of 2^20 boards between two synchronisations: launches enqueued by env2048_step_sequence vs a HIP graph of the same '
                   'launches replayed (one chain / two half-batch chains on two branches); plan of the library at each K: '
                   + str({K: lib.step_sequence_plan(N, K, capi.FLAG_AUTO_RESET) for K in (20, 32, 64)}),
           'rows': rows}
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(json.dumps(out, indent=1) + '\n')


if __name__ == '__main__':
    main()
