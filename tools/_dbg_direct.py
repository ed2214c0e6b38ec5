import ctypes, torch, numpy as np, time, os
from simulate_2048_amd import capi
lib = capi.library()
print(lib.direct_stats(0))
rng = np.random.default_rng(1)
side = [torch.cuda.Stream() for _ in range(3)]   # more HIP streams, as bench.py's other legs create them
for st in side:
    with torch.cuda.stream(st):
        torch.zeros(16, device='cuda').add_(1)
torch.cuda.synchronize()
for n, T in ((1 << 20, 200),):
    b = torch.zeros((n,16), dtype=torch.uint8, device='cuda'); d = torch.zeros((n,), dtype=torch.uint8, device='cuda'); r = torch.zeros((n,), dtype=torch.int32, device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    lib.reset(0, s, n, 0, b.data_ptr(), d.data_ptr(), 1, 0)
    lib.rollout(0, s, n, 0, b.data_ptr(), 1, 0, 64, None, None, None, 3)
    acts = [torch.from_numpy(rng.integers(0, 4, size=n).astype(np.uint8)).cuda() for _ in range(8)]
    table = (ctypes.c_void_p * T)(*[acts[t & 7].data_ptr() for t in range(T)])
    torch.cuda.synchronize()
    for rnd in range(2):
      for flags, name in ((1, 'direct'), (1 | capi.FLAG_NO_DIRECT, 'stream two chains'), (1 | capi.FLAG_NO_DIRECT | capi.FLAG_SINGLE_CHAIN, 'stream one chain')):
        best = 1e9; gpu = 1e9
        for trial in range(9):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lib._dll.env2048_step_sequence(0, s, n, 0, b.data_ptr(), table, r.data_ptr(), d.data_ptr(), None, None, 0, 1, 1 + trial * T, T, flags)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            best = min(best, (t1 - t0) / T * 1e6)
        print(n, T, name, 'wall us/step %.3f' % best)
