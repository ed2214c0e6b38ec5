"""How steady is the two-chain overlap of the direct route?  Distribution of the device-side and wall time of many K-step calls
(2^20 boards, mid-game, host actions).  usage: python tools/diag_direct_jitter.py [K=20] [calls=300]   (ENV2048_LIB=<path>: another build of the library)"""
import ctypes, sys, time, json, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simulate_2048_amd import capi
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lib = capi.Library(os.environ['ENV2048_LIB']) if os.environ.get('ENV2048_LIB') else capi.library(); n = 1 << 20
rng = np.random.default_rng(1)
b = torch.zeros((n, 16), dtype=torch.uint8, device='cuda'); d = torch.zeros((n,), dtype=torch.uint8, device='cuda'); r = torch.zeros((n,), dtype=torch.int32, device='cuda')
s = torch.cuda.current_stream().cuda_stream
lib.reset(0, s, n, 0, b.data_ptr(), d.data_ptr(), 1, 0)
lib.rollout(0, s, n, 0, b.data_ptr(), 1, 0, 64, None, None, None, 3)
acts = [torch.from_numpy(rng.integers(0, 4, size=n).astype(np.uint8)).cuda() for _ in range(8)]
table = (ctypes.c_void_p * K)(*[acts[t & 7].data_ptr() for t in range(K)])
torch.cuda.synchronize()
gpu, wall = [], []
for c in range(calls + 10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lib._dll.env2048_step_sequence(0, s, n, 0, b.data_ptr(), table, r.data_ptr(), d.data_ptr(), None, None, 0, 1, 1 + c * K, K, 1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if c >= 10:
        wall.append((t1 - t0) / K * 1e6); gpu.append(lib.direct_stats(0)['last_gpu_ns'] / K / 1e3)
q = lambda x: [round(float(v), 3) for v in np.percentile(x, [0, 10, 25, 50, 75, 90, 100])]
print(json.dumps({'K': K, 'calls': calls, 'stagger_us': os.environ.get('ENV2048_EXP_DIRECT_STAGGER_US'), 'device_us_per_step_percentiles_0_10_25_50_75_90_100': q(gpu),
                  'wall_us_per_step_percentiles': q(wall), 'reruns': lib.direct_stats(0)['reruns']}))
