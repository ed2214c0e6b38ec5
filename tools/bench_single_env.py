import time, numpy as np, sys
sys.path.insert(0,'/root/repo')
from simulate_2048_amd.twentyfortyeight.envs.twentyfortyeight import TwentyFortyEight
from simulate_2048_amd.twentyfortyeight.core import gameboard
env=TwentyFortyEight(); env.reset(seed=0)
acts=np.random.default_rng(0).integers(0,4,1000)
for a in acts[:50]: env.step(int(a))
t=time.perf_counter(); n=0
for a in acts:
    obs,r,d=env.step(int(a)); n+=1
    if d: env.reset()
dt=time.perf_counter()-t
print('TwentyFortyEight.step: %.0f steps/s (%.1f us/step)'%(n/dt, dt/n*1e6))
b=np.array([[2,2,0,0],[0,4,4,0],[0,0,8,8],[2,0,0,2]])
t=time.perf_counter()
for i in range(300): gameboard.latent_state(b, i&3)
print('latent_state: %.1f us/call'%((time.perf_counter()-t)/300*1e6))
t=time.perf_counter()
for i in range(300): gameboard.next_state(b.copy(), i&3, seed=i)
print('next_state: %.1f us/call'%((time.perf_counter()-t)/300*1e6))
