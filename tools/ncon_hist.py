"""Distribution of contacts / broadphase candidates per mj_step over a random-action rollout (first substep of each env.step)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from robogym_amd.envs.dactyl.locked import make_simple_env
B, T = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 60
env = make_simple_env(batch_size=B, device="cuda:0", starting_seed=3)
env.reset()
sim = env.mujoco_simulation
gen = torch.Generator(device="cuda:0"); gen.manual_seed(0)
off = sim._L.rg_debug_size() - 32 * 8 - 4
hist = torch.zeros(64, dtype=torch.long, device="cuda:0"); hc = torch.zeros(256, dtype=torch.long, device="cuda:0")
orig = sim.env_step
def es(*a, **k):
    if k.get("action") is not None: k["flags"] = 1
    orig(*a, **k)
sim.env_step = es
for t in range(T):
    env.step(torch.rand((B, 20), generator=gen, device="cuda:0") * 2 - 1)
    d = sim.get_field(8)
    hist += torch.bincount(d[:, off].long().clamp(0, 63), minlength=64)
    hc += torch.bincount(d[:, off + 3].long().clamp(0, 255), minlength=256)
h = hist.cpu().numpy(); c = hc.cpu().numpy()
tot = h.sum()
print("ncon histogram (count):", {i: int(v) for i, v in enumerate(h) if v})
cum = h.cumsum() / tot
print("P(ncon<=k):", {k: round(float(cum[k]), 6) for k in (8, 10, 12, 14, 16, 20, 24, 31)})
cc = c.cumsum() / c.sum()
print("ncand: mean %.1f, P(<=64) %.5f P(<=96) %.5f max %d" % ((c * range(256)).sum() / c.sum(), cc[64], cc[96], max(i for i, v in enumerate(c) if v)))
print("status max", int(sim.status.max()))
