"""Bit-identity soak of the substep-granular dispatch (rg_step_items_kernel) against one workgroup per env.step (rg_step_kernel):
two dactyl/locked envs with pipelined resets (same seed: the env kernel's reset draws are counter-based), stepped side by side with
the same random actions for N env.steps -- episodes end by goal time-out and by dropped cubes, every env goes through the reset recipe
many times, contact-rich steps are handed over to the large configuration mid-step -- and compared bit for bit every `--every` steps.

    python tools/soak_items_identity.py [--steps 10000] [--batch 8192] [--every 250]

Exit code 1 at the first difference."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robogym_amd import _native                                                      # noqa: E402
from robogym_amd.envs.dactyl.locked import BatchedLockedEnv, LockedEnvConstants      # noqa: E402
from robogym_amd.mujoco import simulation_interface as si                            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--every", type=int, default=250)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    envs = []
    for which in range(2):
        si.SUBSTEP_ITEMS = bool(which)
        env = BatchedLockedEnv(B, device=dev, constants=LockedEnvConstants(max_timesteps_per_goal=60), starting_seed=5, pipelined_reset=True)
        env.reset()
        envs.append(env)
    state = lambda env: [env.mujoco_simulation.view(f) for f in (_native.RG_F_QPOS, _native.RG_F_QVEL, _native.RG_F_PID, _native.RG_F_WARMSTART, _native.RG_F_CTRL, _native.RG_F_STATUS)]
    gen = torch.Generator(device=dev); gen.manual_seed(9)
    ended = torch.zeros((), dtype=torch.int64, device=dev); started = torch.zeros((), dtype=torch.int64, device=dev)
    redo = [torch.zeros((), dtype=torch.int64, device=dev) for _ in range(2)]
    seen = torch.zeros(B, dtype=torch.int32, device=dev)
    t0 = time.time()
    for k in range(args.steps):
        a = torch.rand((B, 20), generator=gen, device=dev) * 2 - 1
        outs = []
        for which, env in enumerate(envs):
            si.SUBSTEP_ITEMS = bool(which)
            obs, reward, done, info = env.step(a)
            outs.append((obs, reward, done, info))
            if env.mujoco_simulation._redo is not None:
                redo[which] += (env.mujoco_simulation._redo != 0).sum()
        ended += outs[0][2].sum(); started += outs[0][3]["episode_started"].sum()
        seen |= envs[1].sim_status()
        if (k + 1) % args.every == 0 or k + 1 == args.steps:
            same = all(torch.equal(x, y) for x, y in zip(state(envs[0]), state(envs[1])))
            same = same and all(torch.equal(outs[0][0][key], outs[1][0][key]) for key in outs[0][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
            print("step %6d: identical %s; episodes ended %d, started %d; hand-overs %d / %d; status bits seen %d; %.0f s" % (
                k + 1, same, int(ended), int(started), int(redo[0]), int(redo[1]), int(seen.max()), time.time() - t0), flush=True)
            if not same:
                bad = (state(envs[0])[0] != state(envs[1])[0]).any(dim=1).nonzero().flatten()
                print("FIRST DIFFERENCE by step %d: %d envs differ in qpos, e.g. env %d" % (k + 1, len(bad), int(bad[0]) if len(bad) else -1))
                sys.exit(1)
    print("substep-granular dispatch == one workgroup per env.step, bit for bit (state rows, observations, rewards, done), over %d env.steps x %d envs: "
          "%d episodes ended, %d started through the pipelined reset recipe, %d / %d env.steps handed over to the large configuration" % (
              args.steps, B, int(ended), int(started), int(redo[0]), int(redo[1])))


if __name__ == "__main__":
    main()
