import ctypes, sys, time, statistics
mode = sys.argv[1]
hip = ctypes.CDLL('libamdhip64.so')
import torch
if mode != 'auto':
    flag = {'spin': 1, 'yield': 2, 'block': 4}[mode]
    print('hipSetDeviceFlags rc', hip.hipSetDeviceFlags(flag))
sys.path.insert(0, '/root/repo')
from simulate_2048_amd import BatchedEnv2048
x = torch.zeros(8, device='cuda')
torch.cuda.synchronize()
def med(fn, reps=2000):
    w = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); w.append((time.perf_counter() - t0) * 1e6)
    return round(statistics.median(w), 2)
def k_sync():
    x.add_(1); torch.cuda.synchronize()
def k_item():
    x.add_(1); x.cpu()
print(mode, 'tiny kernel + synchronize us', med(k_sync), ' tiny kernel + .cpu() us', med(k_item))
env = BatchedEnv2048(1 << 20, 7, device='cuda', auto_reset=True); env.reset(); env.rollout(64)
acts = torch.randint(0, 4, (20, 1 << 20), dtype=torch.uint8, device='cuda')
def blk():
    env.step_many(acts, out_reward=rw, out_done=dn, mode='launches'); torch.cuda.synchronize()
rw = torch.empty((20, 1 << 20), dtype=torch.int32, device='cuda'); dn = torch.empty((20, 1 << 20), dtype=torch.uint8, device='cuda')
for _ in range(20): blk()
print(mode, 'K=20 block us/step', round(med(blk, 300) / 20, 3))
