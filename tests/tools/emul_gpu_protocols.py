"""The GPU tests' own protocols -- full 40 + 40 mj_steps per env.step -- on the CPU emulation harness (tests/emul: the kernel source compiled for the host).  Used
when a kernel change has to be judged without a GPU call (profiles/r04_emul_gpu_protocols.txt was produced with it).  One env.step of the rearrange worlds
takes ~17 s per env there (ycb ~45 s), so these are minutes-long runs, not part of the CPU suite.

    python tests/tools/emul_gpu_protocols.py resync [nsteps]      tests/test_rearrange_kernel.py::_resync_env_steps(n_substeps=40)
    python tests/tools/emul_gpu_protocols.py ycb_resync           tests/test_rearrange_ycb.py::_resync(40, 10)
    python tests/tools/emul_gpu_protocols.py env                  tests/test_rearrange_env.py::_check_steps(B=2, n_substeps=40, nsteps=12) + goal / tracker checks
    python tests/tools/emul_gpu_protocols.py impulse a|b          the reference's impulse-response pin on BatchedBlockRearrangeEnv (case a: reset on, b: off)
    python tests/tools/emul_gpu_protocols.py runs_clean [ycb]     reset recipe + random-action steps of the batched env, B = 2, status bits / finiteness
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rg_oracle                     # noqa: E402
from robogym_amd import _native                  # noqa: E402


def main():
    which = sys.argv[1]
    rg_oracle.build()
    lib = _native.bind(os.path.join(ROOT, "tests", "emul", "librgstep_emul.so"))
    t0 = time.time()
    np.set_printoptions(precision=2, linewidth=220)
    if which == "resync":
        import tests.test_rearrange_kernel as K
        from robogym_amd.envs.rearrange.xml import load_blocks_model, load_solver_model

        errs = K._resync_env_steps((load_blocks_model(5), load_solver_model()), lib, "cpu", n_substeps=40, nsteps=int(sys.argv[2]) if len(sys.argv) > 2 else 25)
        print(errs); print("median", np.median(errs, axis=0)); print("max", errs.max(axis=0))
        K._assert_resync(errs)
    elif which == "ycb_resync":
        import tests.test_rearrange_ycb as Y
        from robogym_amd.envs.rearrange.xml import load_solver_model, load_ycb_model

        errs = Y._resync((load_ycb_model(8), load_solver_model()), lib, "cpu", 40, 10)
        print("ycb resync med", np.median(errs, axis=0), "max", errs.max(axis=0))
    elif which == "env":
        import tests.test_rearrange_env as E

        env = E._check_steps(lib, "cpu", B=2, n_substeps=40, nsteps=12, tol_scale=3.0)
        print("CHECK_STEPS_OK %.0f s" % (time.time() - t0), flush=True)
        E._goal_and_tracker_checks(env)
    elif which == "impulse":
        import tests.test_rearrange_env as E

        case = {"a": (True, 0.165, 0.036, 5, 1e-3), "b": (False, 0.05, 0.0363, 12, 1.5e-3)}[sys.argv[2]]
        out = E._impulse_response_on_the_kernel(lib, "cpu", stabilize_steps=3, cases=(case,))
        print("case", case, "steady-state displacement x, y, z:", out[0])
    elif which == "runs_clean":
        ycb = len(sys.argv) > 2 and sys.argv[2] == "ycb"
        if ycb:
            from robogym_amd.envs.rearrange.ycb import make_env

            env = make_env(batch_size=2, device="cpu", lib=lib, starting_seed=7, stabilize_steps=5, n_random_initial_steps=2, settle_steps=5)
        else:
            from robogym_amd.envs.rearrange.blocks import BatchedBlockRearrangeEnv

            env = BatchedBlockRearrangeEnv(2, device="cpu", lib=lib, stabilize_steps=5, n_random_initial_steps=2, settle_steps=5)
        env.reset()
        gen = torch.Generator().manual_seed(1)
        for k in range(6 if ycb else 8):
            a = torch.randint(0, 11, (2, 6), generator=gen) if ycb else torch.rand(2, 6, generator=gen) * 2 - 1
            obs, rew, done, info = env.step(a)
            print(k, "done", done.tolist(), "status", int(env.sim.status.max()), int(env.solver_sim.status.max()), "%.0f s" % (time.time() - t0), flush=True)
        env.sync()
        assert int(env.sim.status.max()) == 0 and int(env.solver_sim.status.max()) == 0 and bool(torch.isfinite(env.packed).all())
    else:
        raise SystemExit(__doc__)
    print("OK %s, %.0f s" % (which, time.time() - t0))


if __name__ == "__main__":
    main()
