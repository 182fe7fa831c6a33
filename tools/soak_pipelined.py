"""Long soak with pipelined resets: B envs x T env.steps of random actions, episodes ending by goal time-out (60 steps),
every env going through the reset recipe many times.  Reports status bits, non-finite rows, on-palm rate at episode start."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robogym_amd.envs.dactyl.locked import BatchedLockedEnv, LockedEnvConstants  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda:0")
env = BatchedLockedEnv(B, device=dev, constants=LockedEnvConstants(max_timesteps_per_goal=int(sys.argv[3]) if len(sys.argv) > 3 else 60), starting_seed=5, pipelined_reset=True)
env.reset()
env.mujoco_simulation.set_field(6, torch.zeros((B, 1), dtype=torch.int32, device=dev))
gen = torch.Generator(device=dev); gen.manual_seed(9)
started = torch.zeros((), device=dev); onpalm = torch.zeros((), device=dev); ended = torch.zeros((), device=dev); bad = torch.zeros((), device=dev)
seen = torch.zeros(B, dtype=torch.int32, device=dev)
redo = torch.zeros((), dtype=torch.int64, device=dev)
t0 = time.time()
for t in range(T):
    obs, reward, done, info = env.step(torch.rand((B, 20), generator=gen, device=dev) * 2 - 1)
    s = info["episode_started"]
    started += s.sum(); ended += done.sum()
    onpalm += (s & (0.2 + obs["cube_pos"][:, 2] > 0.04)).sum()
    bad += (~torch.isfinite(obs["qpos"]).all(1)).sum()
    seen |= env.sim_status()
    if env.mujoco_simulation._redo is not None:
        redo += env.mujoco_simulation._redo.sum()
torch.cuda.synchronize(); el = time.time() - t0
st = seen
print("%d envs x %d steps in %.1f s = %.0f env-steps/s (recipe steps included)" % (B, T, el, B * T / el))
print("episodes ended %d, started %d, cube on palm at start %.4f" % (int(ended), int(started), float(onpalm / started.clamp(min=1))))
print("status bits ever seen (envs):", {int(b): int((st & b != 0).sum()) for b in (1, 2, 4, 8, 16, 32)}, " non-finite qpos rows summed over steps:", int(bad))
print("env.steps handed from the rollout to the large kernel configuration (contact / candidate capacity exceeded): %d of %d" % (int(redo), B * T))
