import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from simulate_2048_amd.twentyfortyeight.envs.twentyfortyeight import TwentyFortyEight
from simulate_2048_amd.twentyfortyeight.core import gameboard, gamemove
from simulate_2048_amd.twentyfortyeight.utils.binary import encode_flatten
from simulate_2048_amd.reinforce.game import env as genv, core as gcore
def t(fn, reps=300):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6
b = np.array([[2,2,0,0],[0,4,4,0],[0,0,8,8],[2,0,0,2]])
e = TwentyFortyEight(encoded=True); e.reset(seed=1)
i = [0]
def stepenc():
    i[0] += 1
    o, r, d = e.step(i[0] & 3)
    if d: e.reset()
print('TwentyFortyEight(encoded=True).step %.1f us' % t(stepenc))
print('encode_flatten %.1f us' % t(lambda: encode_flatten(b, 31)))
print('after_state %.1f us' % t(lambda: gameboard.after_state(b)))
print('legal_actions %.1f us' % t(lambda: gamemove.legal_actions(b)))
print('is_done %.1f us' % t(lambda: gameboard.is_done(b)))
print('fill_cells %.1f us' % t(lambda: gameboard.fill_cells(b.copy(), 1, seed=3)))
print('merge_column %.1f us' % t(lambda: gameboard.merge_column(np.array([2,2,4,4]))))
env = genv.Environment(seed=3)
def envstep():
    s, r, d, info = env.step_env(i[0] & 3); i[0] += 1
    if d: env.reset_env()
print('Environment.step_env %.1f us' % t(envstep))
print('Environment.observation %.1f us' % t(lambda: np.array(env.observation())))
st = genv.reset(5)
import torch
bt = torch.tensor(b, dtype=torch.int32).cuda()
print('core.latent_state %.1f us' % t(lambda: gcore.latent_state(bt, 1)))
print('core.slide_and_merge %.1f us' % t(lambda: gcore.slide_and_merge(bt)))
print('core.fill_cells %.1f us' % t(lambda: gcore.fill_cells(bt, 7)))
print('core.create_initial_board %.1f us' % t(lambda: gcore.create_initial_board(7)))
print('core.count_empty %.1f us' % t(lambda: int(gcore.count_empty(bt))))
print('core.merge_row %.1f us' % t(lambda: gcore.merge_row(torch.tensor([2,2,4,4]))))
print('core.max_tile(plain) %.1f us' % t(lambda: int(gcore.max_tile(bt))))
print('core.encode_observation(plain) %.1f us' % t(lambda: np.array(gcore.encode_observation(bt))))
print('core.next_state(plain numpy board) %.1f us' % t(lambda: gcore.next_state(b, 1, 5)))
print('genv.batched_get_observation(plain) %.1f us' % t(lambda: genv.get_observation(genv.GameState(bt, torch.zeros((), dtype=torch.int32).cuda(), torch.zeros((), dtype=torch.bool).cuda(), torch.zeros(()).cuda()))))
