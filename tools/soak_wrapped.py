"""Soak of the reference's default env (make_env(): wrapper stack, randomize=True) with the reference's episode protocol:
every env whose step returned done is reset (reset(mask=done): randomizations redrawn for exactly those envs, the reset recipe
run with them).  Reports status bits, non-finite observations, the spread of the randomized parameters and the throughput of
this synchronous-reset protocol.      python tools/soak_wrapped.py [B] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robogym_amd.envs.dactyl.locked import make_env  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
auto = len(sys.argv) > 3 and sys.argv[3] == "auto"      # pipelined in-step resets + wrapper auto_reset instead of reset(mask=done)
env = make_env(batch_size=B, device="cuda:0", starting_seed=7, constants={"max_timesteps_per_goal": 60}, pipelined_reset=auto)
gen = torch.Generator(device="cuda:0"); gen.manual_seed(1)
obs = env.reset()
P = env.unwrapped.mujoco_simulation.params
ends = resets = bad = drops = 0
status_seen = 0
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(steps):
    obs, reward, done, info = env.step(torch.randint(0, 11, (B, 20), generator=gen, device="cuda:0"))
    if auto:      # no host synchronisation inside the loop
        bad_t = sum((~torch.isfinite(v.float())).sum() for v in obs.values()) + (0 if t == 0 else bad_t)
        st_t = env.unwrapped.sim_status().max() if t == 0 else torch.maximum(st_t, env.unwrapped.sim_status().max())
        drops_t = (reward[:, 3] < 0).sum() + (0 if t == 0 else drops_t)
    else:
        bad += int(sum((~torch.isfinite(v.float())).sum() for v in obs.values()))
        status_seen |= int(env.unwrapped.sim_status().max())
        drops += int((reward[:, 3] < 0).sum())
    if auto:
        ends_t = done.to(torch.int32) if t == 0 else ends_t + done.to(torch.int32)
        continue
    n = int(done.sum())
    if n:
        ends += n; resets += 1
        env.reset(done)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
if auto:
    ends, bad, status_seen, drops = int(ends_t.sum()), int(bad_t), int(st_t), int(drops_t)
print("make_env(%s) [randomize=True], B=%d, %d steps with %s: %.1f s = %.0f env-steps/s" % ("pipelined_reset=True" if auto else "", B, steps, "in-step resets (wrapper auto_reset)" if auto else "reset(mask=done) after every step that ended episodes", dt, B * steps / dt))
print("episodes ended %d (%d with the drop penalty) in %d reset calls; non-finite observation entries %d; status bits ever seen 0x%x" % (ends, drops, resets, bad, status_seen))
print("parameter spread over the batch now: gravity z %.2f..%.2f | cube size factor %.3f..%.3f | timestep %.4f..%.4f | kp[0] %.3f..%.3f | cube sliding friction %.2f..%.2f" % (
    float(P["gravity"][:, 2].min()), float(P["gravity"][:, 2].max()), float(P["geom_scale"].min()), float(P["geom_scale"].max()), float(P["timestep"].min()), float(P["timestep"].max()),
    float(P["actuator_gainprm"][:, 0, 0].min()), float(P["actuator_gainprm"][:, 0, 0].max()), float(P["geom_friction"][:, 0, 0].min()), float(P["geom_friction"][:, 0, 0].max())))
