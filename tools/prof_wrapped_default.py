"""The default make_env() (wrapper stack, randomize=True) for a kernel trace: which of the wrapper stack's tensor kernels carry its ~1 ms per step.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_wrapped -- python tools/prof_wrapped_default.py [B] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robogym_amd.envs.dactyl.locked import make_env  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
gen = torch.Generator(device="cuda:0"); gen.manual_seed(0)
env = make_env(batch_size=B, device="cuda:0", starting_seed=1)
env.reset()
for _ in range(steps):
    env.step(torch.randint(0, 11, (B, 20), generator=gen, device="cuda:0"))
torch.cuda.synchronize()
print("steps", steps)
