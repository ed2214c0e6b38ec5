import sys, time, json
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
import numpy as np, torch
from simulate_2048_amd import BatchedEnv2048, capi, runtime
from simulate_2048_amd.reinforce.game import env as genv
from simulate_2048_amd.trajectory import TrajectoryBuffer

def per_call(fn, reps):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6

out = {}
for nn in (8, 1 << 20):
    keys_np = np.stack([np.zeros(nn, dtype=np.uint32), np.arange(nn, dtype=np.uint32)], axis=1)
    keys_dev = torch.from_numpy(keys_np.view(np.int32)).cuda()
    st = genv.batched_reset(keys_np)
    acts = torch.randint(0, 4, (nn,), dtype=torch.int32).cuda()
    out[f'step_np_keys_{nn}'] = per_call(lambda: genv.batched_step(st, acts, keys_np), 100)
    out[f'step_dev_keys_{nn}'] = per_call(lambda: genv.batched_step(st, acts, keys_dev), 100)
    out[f'obs_{nn}'] = per_call(lambda: genv.batched_get_observation(st), 100)
    out[f'legal_{nn}'] = per_call(lambda: genv.batched_get_legal_actions(st), 100)
    plain = genv.GameState(*[x.as_subclass(torch.Tensor) for x in st])
    out[f'legal_plain_input_{nn}'] = per_call(lambda: genv.batched_get_legal_actions(plain), 100)
    rt = runtime.current()
    boards = plain.board.reshape(-1, 16)
    legal = torch.empty((nn, 4), dtype=torch.uint8, device='cuda')
    dev, stream = rt.handles()
    out[f'legal_kernel_only_{nn}'] = per_call(lambda: rt.lib.status_i32(dev, stream, nn, boards.data_ptr(), None, legal.data_ptr(), None), 100)
n = 1 << 20
e = BatchedEnv2048(n, 4, device='cuda:0', auto_reset=True); e.reset()
T = 256
traj = TrajectoryBuffer(e, T, record_obs=False, record_legal=False); traj.begin()
for t in range(T): traj.step(None)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); ep = traj.episodes(); torch.cuda.synchronize(); out[f'episodes_call_{rep}_ms'] = (time.perf_counter() - t0) * 1e3
dev, stream = 0, torch.cuda.current_stream().cuda_stream
counts = torch.zeros((n,), dtype=torch.int32, device='cuda')
out['count_kernel_us'] = per_call(lambda: e.lib.count_episodes(dev, stream, n, n, T, 0, T, traj.done.data_ptr(), 0, counts.data_ptr()), 5)
ends = torch.cumsum(counts.to(torch.int64), 0); offsets = (ends - counts).contiguous(); total = int(ends[-1])
raw = torch.zeros((total, 24), dtype=torch.uint8, device='cuda')
out['segment_kernel_us'] = per_call(lambda: e.lib.segment_episodes(dev, stream, n, n, T, 0, T, traj.done.data_ptr(), traj.reward.data_ptr(), traj.max_exp.data_ptr(), 0, offsets.data_ptr(), raw.data_ptr(), None), 5)
print(json.dumps(out, indent=1))
