"""The direct route of env2048_step_sequence against the stream route over batch sizes (where does keeping the boards in the XCD L2s
stop paying?  profiles/r05_z_direct_route_size_limit.jsonl; ENV2048_EXP_DIRECT_MAX_LOG2=23 lets the route take sizes beyond its limit)."""
import ctypes, sys, time, json, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simulate_2048_amd import capi
lib = capi.library()
rng = np.random.default_rng(1)
for n, K in ((1 << 19, 100), (3 << 18, 100), (1 << 20, 100), (5 << 18, 100), (3 << 19, 100), (7 << 18, 100), (1 << 21, 100)):
    logn = round(float(np.log2(n)), 2)
    b = torch.zeros((n, 16), dtype=torch.uint8, device='cuda'); d = torch.zeros((n,), dtype=torch.uint8, device='cuda'); r = torch.zeros((n,), dtype=torch.int32, device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    lib.reset(0, s, n, 0, b.data_ptr(), d.data_ptr(), 1, 0)
    lib.rollout(0, s, n, 0, b.data_ptr(), 1, 0, 64, None, None, None, 3)
    acts = [torch.from_numpy(rng.integers(0, 4, size=n).astype(np.uint8)).cuda() for _ in range(4)]
    table = (ctypes.c_void_p * K)(*[acts[t & 3].data_ptr() for t in range(K)])
    torch.cuda.synchronize()
    res = {}
    for name, flags in (('default', 1), ('stream', 1 | capi.FLAG_NO_DIRECT)):
        best = 1e9
        for trial in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            lib._dll.env2048_step_sequence(0, s, n, 0, b.data_ptr(), table, r.data_ptr(), d.data_ptr(), None, None, 0, 1, 1 + trial * K, K, flags)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / K * 1e6)
        res[name] = round(best, 2)
    print(json.dumps({'log2_boards': logn, 'K': K, 'us_per_step': res, 'direct_calls': lib.direct_stats(0)['calls']}))
