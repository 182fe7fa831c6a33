import sys; sys.path.insert(0,'.')
import torch, numpy as np
from robogym_amd.envs.dactyl.locked import LockedSimulation, load_locked_model
B=512
model=load_locked_model()
for fa,fb in ((0,4),(0,8),(4,12)):
    sims=[LockedSimulation(model,B,device='cuda:0') for _ in range(2)]
    gen=torch.Generator(device='cuda:0'); gen.manual_seed(5)
    first=None
    for k in range(40):
        a=torch.rand((B,20),generator=gen,device='cuda:0')*2-1 if k>=10 else torch.zeros((B,20),device='cuda:0')
        for sim,fl in zip(sims,(fa,fb)): sim.env_step(action=a,nforward_ticks=3,flags=fl)
        d=(sims[0].qpos!=sims[1].qpos).any(1)
        if d.any() and first is None:
            first=k; print('flags',fa,fb,'first diff at step',k,'envs',d.nonzero().flatten().tolist()[:10], 'maxdiff',(sims[0].qpos-sims[1].qpos).abs().max().item())
    print('flags',fa,fb,'final differing envs',int(d.sum()), 'status', int(sims[0].status.max()), int(sims[1].status.max()))
