"""rearrange/ycb (BASELINE.json configs[4]: `make_env(parameters={"simulation_params": {"num_objects": 8}})`), batched on the MI355X.

The reference env (/root/reference/robogym/envs/rearrange/ycb.py:46-96, common/mesh.py:48-110, simulation/mesh.py:43-70) is the rearrange
env of envs/rearrange/blocks.py with mesh objects: every object is one free body whose geoms are the convex parts of a YCB scan.  Everything
around the objects -- TCP-controlled UR16e with its solver simulation, the 40 + 40 mj_steps per env.step, observation row, reward, goal
tracker, contact scans, safety stop -- is the same code here as there, so this env IS `BatchedBlockRearrangeEnv` on another compiled model;
the stepper runs it on `rb_step_kernel`'s medium configuration (one wave per env, 56 dofs, 10 envs per CU).

Built: the world with a FIXED set of objects per compiled model (`load_ycb_model`: the reference's `_sample_object_meshes` draw with seed 0).
Not built: a new object set per episode and env (the reference re-creates the simulation at every reset,
common/base.py:850-856); `normalize_mesh`, object-scale randomisation, the mesh envs' damping change while objects settle."""
from robogym_amd.envs.rearrange.blocks import BatchedBlockRearrangeEnv
from robogym_amd.envs.rearrange.xml import load_ycb_model


class BatchedYcbRearrangeEnv(BatchedBlockRearrangeEnv):
    def __init__(self, batch_size: int, device="cuda:0", num_objects: int = 8, **kw):
        model = kw.pop("main_model", None) or load_ycb_model(num_objects)
        super().__init__(batch_size, device=device, num_objects=num_objects, main_model=model, **kw)
        self.object_names = list(model.names.get("object_mesh", []))

    def info(self):
        out = super().info()
        out["object_names"] = self.object_names       # YcbRearrangeEnv._get_simulation_info (ycb.py:86-90): the mesh name of every object
        return out


class GroupedYcbRearrangeEnv:
    """rearrange/ycb with DIFFERENT object sets across the batch: the envs are split into equal groups, every group runs its own compiled model (one of the
    shipped object sets, `xml.YCB_SHIPPED_SETS`) as a `BatchedYcbRearrangeEnv` on a stream of its own, so the groups' launches overlap on the GPU; `step` / `reset` /
    `observe` speak for the whole batch (rows in group order).  This is per-ENV variety with a fixed set per env slot -- the reference draws a new set per EPISODE and
    rebuilds the simulation (envs/rearrange/ycb.py:58-84, common/base.py:850-856), which needs per-env geometry rows in the stepper."""

    def __init__(self, batch_size: int, device="cuda:0", object_sets=(0, 1, 2, 3), starting_seed: int = 0, **kw):
        import torch

        K = len(object_sets)
        assert batch_size % K == 0, "batch_size must be a multiple of the number of object sets"
        self.B, self.K, self.b = int(batch_size), K, int(batch_size) // K
        self.groups = [BatchedYcbRearrangeEnv(self.b, device=device, main_model=load_ycb_model(kw.get("num_objects", 8), set_index=int(k)), starting_seed=starting_seed + 1000 * i, **kw)
                       for i, k in enumerate(object_sets)]
        g0 = self.groups[0]
        self.device, self.N, self.obs_dim, self.wrapped, self.action_shape = g0.device, g0.N, g0.obs_dim, g0.wrapped, (self.B, 6)
        self._cuda = self.device.type == "cuda"
        self.streams = [torch.cuda.Stream(self.device) for _ in self.groups] if self._cuda else [None] * K
        self.object_names = [g.object_names for g in self.groups]

    def _each(self, fn):
        """fn(group index, group) for every group, on the group's stream; the caller's stream waits for all of them afterwards"""
        import torch

        out = []
        cur = torch.cuda.current_stream(self.device) if self._cuda else None
        for i, (g, st) in enumerate(zip(self.groups, self.streams)):
            if st is None:
                out.append(fn(i, g))
            else:
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    out.append(fn(i, g))
        if cur is not None:
            for st in self.streams:
                cur.wait_stream(st)
        return out

    def _cat(self, outs):
        import torch

        obs = {k: torch.cat([o[0][k] for o in outs]) for k in outs[0][0]}
        info = {k: (torch.cat([o[3][k] for o in outs]) if torch.is_tensor(outs[0][3][k]) else [x for o in outs for x in [o[3][k]] * self.b]) for k in outs[0][3]}
        return obs, torch.cat([o[1] for o in outs]), torch.cat([o[2] for o in outs]), info

    def reset(self, mask=None):
        import torch

        outs = self._each(lambda i, g: g.reset(None if mask is None else mask[i * self.b:(i + 1) * self.b]))
        return {k: torch.cat([o[k] for o in outs]) for k in outs[0]}

    def step(self, actions):
        b = self.b
        self._each(lambda i, g: g._step_launch(actions[i * b:(i + 1) * b].contiguous()))       # every group's three launches are queued ...
        return self._cat(self._each(lambda i, g: g._step_finish()))                             # ... before any group's flags are read back

    def observe(self):
        import torch

        outs = [g.observe() for g in self.groups]
        return {k: torch.cat([o[k] for o in outs]) for k in outs[0]}

    def sync(self):
        for g in self.groups:
            g.sync()

    def status(self):
        import torch

        return torch.cat([torch.maximum(g.sim.status.reshape(-1), g.solver_sim.status.reshape(-1)) for g in self.groups])


def make_env(batch_size: int = 4096, device="cuda:0", parameters=None, constants=None, starting_seed: int = 0, apply_wrappers: bool = True, **kw):
    """`YcbRearrangeEnv.build` surface (ycb.py:96) for the batched env; accepts what envs/rearrange/blocks.py `make_env` accepts."""
    parameters, constants = dict(parameters or {}), dict(constants or {})
    sp, rc = dict(parameters.get("simulation_params", {})), dict(parameters.get("robot_control_params", {}))
    if parameters.get("mesh_names") is not None or constants.get("normalize_mesh"):
        raise NotImplementedError("mesh_names / normalize_mesh: the shipped model holds one fixed object set")
    args = dict(num_objects=sp.get("num_objects", 8), max_position_change=rc.get("max_position_change", 0.1), arm_reset_controller_error=rc.get("arm_reset_controller_error", True),
                n_random_initial_steps=parameters.get("n_random_initial_steps", 10), starting_seed=starting_seed, wrappers=bool(apply_wrappers),
                n_action_bins=constants.get("n_action_bins", 11))
    for k in ("success_threshold", "successes_needed", "success_reward", "max_timesteps_per_goal_per_obj"):
        if k in constants:
            args[k] = constants[k]
    args.update(kw)
    object_sets = args.pop("object_sets", None)      # e.g. (0, 1, 2, 3): different object sets across the batch (GroupedYcbRearrangeEnv)
    if object_sets is not None and len(object_sets) > 1:
        return GroupedYcbRearrangeEnv(batch_size, device=device, object_sets=tuple(object_sets), **args)
    return BatchedYcbRearrangeEnv(batch_size, device=device, **args)


def make_simple_env(*a, **kw):
    kw["apply_wrappers"] = False
    return make_env(*a, **kw)
