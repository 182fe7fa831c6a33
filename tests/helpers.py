"""Shared helpers for the parity tests: drive the oracle and a kernel build from identical bytes."""
import numpy as np
import torch

NON_TARGET_QPOS = [i for i in range(38) if not (7 <= i < 14)]  # the target cube free-falls, excluded


def settle_oracle(env_oracle, nsteps, ctrl=None):
    if ctrl is not None:
        env_oracle.sim.ctrl[:] = ctrl
    env_oracle.settle(nsteps)


def sync_state_from_oracle(sim, env_oracle, rows=None):
    """kernel state <- oracle state rounded to fp32; oracle <- the same rounded values."""
    st = env_oracle.get_state_f32()
    env_oracle.set_state_f32(st)
    B = sim.batch_size
    full = sim.get_state()
    for k, v in st.items():
        t = torch.tensor(v, device=sim.device)
        if rows is None:
            full[k][:] = t
        else:
            full[k][rows] = t
    sim.set_state({k: full[k] for k in st})


def resync_errors(sim, env_oracle, actions, substep_level=False, detail=False):
    """Re-synchronised errors: the kernel is restarted from the oracle's (fp32-rounded) state before
    every env.step (10 substeps + 3 forward ticks) or, with `substep_level`, before every mj_step.
    Returns rows of (qpos Linf over non-target joints, qvel Linf, pid Linf); `detail` appends (largest contact count of the
    env.step's substeps, deepest penetration) for the tail characterisation."""
    out = []
    for a in actions:
        if substep_level:
            sync_state_from_oracle(sim, env_oracle)
            ctrl = env_oracle.denormalize(np.clip(a, -1, 1), env_oracle.relative_action)
            for _ in range(env_oracle.n_substeps):
                env_oracle.sim.ctrl[:] = ctrl
                sync_state_from_oracle(sim, env_oracle)
                sim.env_step(nsubsteps=1, nforward_ticks=0)
                env_oracle.sim.step()
                q = sim.qpos.cpu().numpy()[0].astype(np.float64)
                v = sim.qvel.cpu().numpy()[0].astype(np.float64)
                p = sim.get_field(3).cpu().numpy()[0].astype(np.float64)
                out.append((np.abs(q - env_oracle.sim.qpos)[NON_TARGET_QPOS].max(), np.abs(v - env_oracle.sim.qvel).max(), np.abs(p - env_oracle.sim.pid).max()))
            continue
        sync_state_from_oracle(sim, env_oracle)
        at = torch.tensor(np.repeat(a[None].astype(np.float32), sim.batch_size, 0), device=sim.device)
        sim.env_step(action=at, nforward_ticks=3)
        extra = ()
        if detail:   # the oracle substep by substep (same arithmetic as env_step), recording what the step went through
            o = env_oracle
            o.sim.ctrl[:] = o.denormalize(np.clip(np.asarray(a, dtype=np.float64), -1, 1), o.relative_action)
            ncon_max, depth = 0, 0.0
            for _ in range(o.n_substeps):
                o.sim.step()
                ncon_max = max(ncon_max, o.sim.ncon)
                for c in o.sim.contacts():
                    depth = max(depth, -c["dist"])
            o.sim.forward(); o.sim.forward()
            dist = o.goal_distance(); o.prev_dist = dist
            o.sim.forward()
            extra = (ncon_max, depth)
        else:
            env_oracle.env_step(a)
        q = sim.qpos.cpu().numpy()[0].astype(np.float64)
        v = sim.qvel.cpu().numpy()[0].astype(np.float64)
        p = sim.get_field(3).cpu().numpy()[0].astype(np.float64)
        out.append((np.abs(q - env_oracle.sim.qpos)[NON_TARGET_QPOS].max(), np.abs(v - env_oracle.sim.qvel).max(), np.abs(p - env_oracle.sim.pid).max()) + extra)
    return np.array(out)
