import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from robogym_amd.envs.dactyl.locked import make_env
B=8192
env = make_env(batch_size=B, device="cuda:0", starting_seed=7, constants={"max_timesteps_per_goal": 60}, pipelined_reset=True)
gen = torch.Generator(device="cuda:0"); gen.manual_seed(1)
env.reset()
acc = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = f(*a, **k); e.record()
        acc.setdefault(name, []).append((s, e)); return r
    setattr(obj, name, g)
for n in ("_randomize_before_reset", "_after_env_step", "_observation", "_episode_start", "_action_noise_reset", "_backlash"):
    wrap(env, n)
wrap(env.env, "step")
wrap(env.env.mujoco_simulation, "set_constants")
wrap(env.env.mujoco_simulation, "env_step")
for _ in range(100): env.step(torch.randint(0, 11, (B, 20), generator=gen, device="cuda:0"))
acc.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 100
for _ in range(N): env.step(torch.randint(0, 11, (B, 20), generator=gen, device="cuda:0"))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("per step %.2f ms" % (1e3 * dt / N))
for k, v in acc.items():
    print("  %-26s %.2f ms per step (%d calls)" % (k, sum(a.elapsed_time(b) for a, b in v) / N, len(v)))
