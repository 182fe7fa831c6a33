"""Soak run: B envs x T env.steps of random actions through the full env (resets of done envs included);
reports status bits, non-finite states, episode statistics.   python tools/soak.py [B] [T]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robogym_amd.envs.dactyl.locked import make_simple_env  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
env = make_simple_env(batch_size=B, device=dev, starting_seed=123)
t0 = time.time(); env.reset(); torch.cuda.synchronize(); t_reset = time.time() - t0
st0 = env.sim_status()
print("reset of %d envs: %.2f s; status bits after reset: %s" % (B, t_reset, {int(b): int((st0 & b != 0).sum()) for b in (1, 2, 4, 8, 16)}))
env.mujoco_simulation.set_field(6, torch.zeros((B, 1), dtype=torch.int32, device=dev))
gen = torch.Generator(device=dev); gen.manual_seed(1)
ndone = 0; nsucc = 0; t0 = time.time(); t_masked = 0.0
for t in range(T):
    a = torch.rand((B, 20), generator=gen, device=dev) * 2 - 1
    obs, reward, done, info = env.step(a)
    nd = int(done.sum().item()); ndone += nd
    if nd:
        torch.cuda.synchronize(); t1 = time.time()
        env.reset(mask=done)
        torch.cuda.synchronize(); t_masked += time.time() - t1
torch.cuda.synchronize(); el = time.time() - t0
st = env.sim_status()
q = env.mujoco_simulation.qpos
print("%d steps: %.1f s total, of which %.1f s in %d masked resets (%d env resets)" % (T, el, t_masked, T, ndone))
print("status bits after rollout:", {int(b): int((st & b != 0).sum()) for b in (1, 2, 4, 8, 16)}, "non-finite qpos rows:", int((~torch.isfinite(q).all(1)).sum()))
print("cube z above palm plane (>0.04 from floor offset) fraction: %.3f" % ((0.2 + obs["cube_pos"][:, 2]) > 0.04).float().mean().item())
