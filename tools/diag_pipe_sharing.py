"""Does the direct route's set of queues slow down the process's own HIP streams?  The stream route's two-chain sequence (two HIP
streams + helper thread) timed in four situations: with / without the library's queues (ENV2048_DIRECT=0 in a child process),
with / without three more HIP streams of the caller's.  usage: python tools/diag_pipe_sharing.py"""
import ctypes, json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = os.environ.get('PIPE_CHILD')
if CHILD is None:
    for direct in ('1', '0'):
        for extra in ('0', '3'):
            env = dict(os.environ, PIPE_CHILD=extra, ENV2048_DIRECT=direct)
            out = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=300)
            print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
    sys.exit(0)

import numpy as np, torch
from simulate_2048_amd import capi
lib = capi.library()
info = lib.direct_stats(0)            # (starts the route's queues when ENV2048_DIRECT != 0: BEFORE any HIP stream of ours exists)
extra = [torch.cuda.Stream() for _ in range(int(CHILD))]
for st in extra:
    with torch.cuda.stream(st):
        torch.zeros(16, device='cuda').add_(1)
torch.cuda.synchronize()
n, T = 1 << 20, 200
rng = np.random.default_rng(1)
b = torch.zeros((n, 16), dtype=torch.uint8, device='cuda'); d = torch.zeros((n,), dtype=torch.uint8, device='cuda'); r = torch.zeros((n,), dtype=torch.int32, device='cuda')
s = torch.cuda.current_stream().cuda_stream
lib.reset(0, s, n, 0, b.data_ptr(), d.data_ptr(), 1, 0)
lib.rollout(0, s, n, 0, b.data_ptr(), 1, 0, 64, None, None, None, 3)
acts = [torch.from_numpy(rng.integers(0, 4, size=n).astype(np.uint8)).cuda() for _ in range(8)]
table = (ctypes.c_void_p * T)(*[acts[t & 7].data_ptr() for t in range(T)])
res = {'library_queues': info['available'], 'extra_hip_streams': int(CHILD)}
for name, flags in (('stream_two_chains', 1 | capi.FLAG_NO_DIRECT), ('stream_one_chain', 1 | capi.FLAG_NO_DIRECT | capi.FLAG_SINGLE_CHAIN)):
    best = 1e9
    for trial in range(9):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lib._dll.env2048_step_sequence(0, s, n, 0, b.data_ptr(), table, r.data_ptr(), d.data_ptr(), None, None, 0, 1, 1 + trial * T, T, flags)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / T * 1e6)
    res[name + '_us_per_step'] = round(best, 3)
print(json.dumps(res))
