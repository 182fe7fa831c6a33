"""rearrange/ycb (BASELINE.json configs[4]: `make_env(parameters={"simulation_params": {"num_objects": 8}})`), batched on the MI355X.

The reference env (/root/reference/robogym/envs/rearrange/ycb.py:46-96, common/mesh.py:48-110, simulation/mesh.py:43-70) is the rearrange
env of envs/rearrange/blocks.py with mesh objects: every object is one free body whose geoms are the convex parts of a YCB scan.  Everything
around the objects -- TCP-controlled UR16e with its solver simulation, the 40 + 40 mj_steps per env.step, observation row, reward, goal
tracker, contact scans, safety stop -- is the same code here as there, so this env IS `BatchedBlockRearrangeEnv` on another compiled model;
the stepper runs it on `rb_step_kernel`'s medium configuration (one wave per env, 56 dofs, 10 envs per CU).

Built: the world with a FIXED set of objects per compiled model (`load_ycb_model`: the reference's `_sample_object_meshes` draw with seed 0 + set index); a new
object set per episode out of the shipped sets by moving envs between the groups' slots (`GroupedYcbRearrangeEnv`).
Not built: a set sampled from the whole YCB catalogue per episode (the reference re-creates the simulation at every reset, common/base.py:850-856:
needs per-env geometry rows in the stepper); `normalize_mesh`, object-scale randomisation, the mesh envs' damping change while objects settle."""
from robogym_amd import _native
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


class _LazyNames:
    """info["object_names"] of a step with device-side slot trading: the per-env list of object names, resolved (one readback) when it is looked at"""

    def __init__(self, names_of_group, slot_dev, b):
        self._g, self._s, self._b, self._v = names_of_group, slot_dev, b, None

    def _list(self):
        if self._v is None:
            self._v = [self._g[int(x) // self._b] for x in self._s.cpu().numpy()]
        return self._v

    def __len__(self):
        return int(self._s.shape[0])

    def __getitem__(self, i):
        return self._list()[i]

    def __iter__(self):
        return iter(self._list())

    def __eq__(self, other):
        return list(self) == list(other)


class GroupedYcbRearrangeEnv:
    """rearrange/ycb with DIFFERENT object sets across the batch and a NEW set per episode.  The batch is split into equal groups, every group runs its own
    compiled model (one of the shipped object sets, `xml.YCB_SHIPPED_SETS`) as a `BatchedYcbRearrangeEnv` on a stream of its own, so the groups' launches
    overlap on the GPU; `step` / `reset` / `observe` speak for the whole batch.

    The reference draws a new object set at every reset and rebuilds the simulation (envs/rearrange/ycb.py:58-84, common/base.py:850-856).  Here the compiled
    models stay where they are and the ENVS move: env i of the batch is served by a physics slot `slot[i]` (a row of one group), and the slots whose episodes end
    on the same step -- all of them are about to run the reset recipe from a freshly made world, nothing of the old episode survives in them -- are handed out
    again among the envs that ended, by a random permutation.  The next episode of env i therefore runs on the object set of whatever group its new slot belongs
    to.  With pipelined resets the slots that ended on EARLIER steps and are still inside the reset recipe take part in the deal as well (they are as ownerless as
    the ones that just ended), so an env that ends draws its next slot from everything that is resetting at that moment -- recipe length x episode ends per step,
    hundreds of slots at B = 4096.  Limits, stated: the pool of object sets is the K shipped ones (the reference samples `num_objects` meshes from the whole YCB
    catalogue); an env that ends while nothing else is resetting keeps its slot.  `resample_object_sets=False` pins env i to slot i."""

    def __init__(self, batch_size: int, device="cuda:0", object_sets=(0, 1, 2, 3), starting_seed: int = 0, resample_object_sets: bool = True, multi_launch: bool = True, **kw):
        import numpy as np
        import torch

        K = len(object_sets)
        # device_reset (the recipe kernel): the slot trading runs on the device too (round 6, _trade_slots_device): no readback of the ended episodes in a step call
        self._device_trade = bool(kw.get("device_reset")) and bool(resample_object_sets)
        assert batch_size % K == 0, "batch_size must be a multiple of the number of object sets"
        self.B, self.K, self.b = int(batch_size), K, int(batch_size) // K
        self.object_sets = tuple(int(k) for k in object_sets)
        self.groups = [BatchedYcbRearrangeEnv(self.b, device=device, main_model=load_ycb_model(kw.get("num_objects", 8), set_index=k), starting_seed=starting_seed + 1000 * i, **kw)
                       for i, k in enumerate(self.object_sets)]
        g0 = self.groups[0]
        # every group's physics phase as ONE launch when the groups' kernels match (same batch size by construction; same one-wave configuration) and there are <= 8
        self.multi_launch = bool(multi_launch) and K <= 8 and len({g.sim.info["threads"] for g in self.groups}) == 1 and len({g.sim.info["lds_bytes"] for g in self.groups}) == 1
        self.device, self.N, self.obs_dim, self.wrapped, self.action_shape = g0.device, g0.N, g0.obs_dim, g0.wrapped, (self.B, g0.action_dim)
        self._cuda = self.device.type == "cuda"
        self.streams = [torch.cuda.Stream(self.device) for _ in self.groups] if self._cuda else [None] * K
        self.object_names = [g.object_names for g in self.groups]
        self.resample = bool(resample_object_sets)
        self._rng = np.random.RandomState(starting_seed + 77)
        self._slot_host = np.arange(self.B)                          # env -> physics slot (group * b + row); with device trading the device table is the authority
        self._slot_dev = torch.arange(self.B, device=self.device)
        self._env_of_slot_dev = torch.arange(self.B, device=self.device)
        self.episodes_moved = 0                                      # episodes that started on another group's object set than the previous one of that env
        if self._device_trade:
            self._gen = torch.Generator(device=self.device); self._gen.manual_seed(starting_seed + 77)
            self._moved_dev = torch.zeros((), dtype=torch.int64, device=self.device)
            self._arange = torch.arange(self.B, device=self.device)

    @property
    def _slot(self):
        """env -> slot as a host array (with device trading: read back from the device table -- tests and reset() only, never inside step())"""
        if getattr(self, "_device_trade", False):
            self._slot_host = self._slot_dev.cpu().numpy()
        return self._slot_host

    @_slot.setter
    def _slot(self, v):
        self._slot_host = v

    # ------------------------------------------------------------------ env <-> slot
    def _reassign(self, envs):
        """The slots of `envs` (all of them at the start of a reset) handed out again among these envs."""
        import numpy as np
        import torch

        if not self.resample or len(envs) < 2:
            return
        table = self._slot.copy()
        old = table[envs]
        new = old[self._rng.permutation(len(envs))]
        self.episodes_moved += int((old // self.b != new // self.b).sum())
        table[envs] = new
        self._slot_host = table
        self._slot_dev = torch.as_tensor(table, device=self.device)
        inv = np.empty(self.B, dtype=np.int64); inv[table] = np.arange(self.B)
        self._env_of_slot_dev = torch.as_tensor(inv, device=self.device)

    def object_set_of_env(self):
        """index into `object_sets` of the set every env of the batch currently has on its table"""
        if self._device_trade:
            self.episodes_moved = int(self._moved_dev)          # (a readback: this is a host query)
        return self._slot // self.b

    def _to_slots(self, x):
        """a per-env tensor in slot order"""
        return x if not self.resample else x[self._env_of_slot_dev]

    def _to_envs(self, x):
        """a per-slot tensor in env order"""
        return x if not self.resample else x[self._slot_dev]

    # ------------------------------------------------------------------ the groups
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

    def _observation(self):
        import torch

        packed = self._to_envs(torch.cat([g.packed for g in self.groups]))
        ema = self._to_envs(torch.cat([g.action_ema for g in self.groups])) if self.wrapped else None
        return self.groups[0].observe(packed, ema)

    def _names(self):
        return [self.object_names[s // self.b] for s in self._slot]

    def reset(self, mask=None):
        import numpy as np

        envs = np.arange(self.B) if mask is None else np.nonzero(mask.cpu().numpy())[0]
        self._reassign(envs)                                          # a reset is a new episode: a new object set with it
        smask = None if mask is None else self._to_slots(mask)
        self._each(lambda i, g: g.reset(None if smask is None else smask[i * self.b:(i + 1) * self.b]))
        return self._observation()

    def step(self, actions):
        import numpy as np
        import torch

        b = self.b
        sact = self._to_slots(actions)
        if self.multi_launch:
            # ONE launch per phase over all groups (rb_multi_begin / rb_multi_launch): the groups' worlds side by side at full occupancy.  (One launch chain per group
            # on its own stream, the alternative below, overlaps little on the GPU: 1.5 chains' worth at four groups, tools/ycb_sets_overlap.py)
            L, st = self.groups[0]._L, self.groups[0]._stream()
            for phase in ("solver", "main"):
                _native.check(L, L.rb_multi_begin(), "rb_multi_begin")
                try:
                    for i, g in enumerate(self.groups):
                        g._step_launch(sact[i * b:(i + 1) * b].contiguous() if phase == "solver" else None, phase=phase)
                finally:
                    _native.check(L, L.rb_multi_launch(st), "rb_multi_launch")
            for g in self.groups:
                g._step_launch(None, phase="post")
            outs = [g._step_finish() for g in self.groups]
        else:
            self._each(lambda i, g: g._step_launch(sact[i * b:(i + 1) * b].contiguous()))           # every group's three launches are queued ...
            outs = self._each(lambda i, g: g._step_finish())                                        # ... before any group's flags are read back
        obs = self._observation()
        reward, done = self._to_envs(torch.cat([o[1] for o in outs])), self._to_envs(torch.cat([o[2] for o in outs]))
        info = {k: self._to_envs(torch.cat([o[3][k] for o in outs])) for k in outs[0][3] if torch.is_tensor(outs[0][3][k])}
        if self._device_trade:
            # the names and the table of THIS step are resolved on request (a readback); nothing in a step call waits for the GPU
            snap = self._slot_dev.clone()
            info["object_names"] = _LazyNames(self.object_names, snap, self.b)
            self._slot_of_step_dev = snap
            self._trade_slots_device()
        else:
            info["object_names"] = self._names()            # (of the episode this step belonged to: the terminal step still names the set that just ended)
            self._slot_of_step = self._slot.copy()
            self._trade_slots()
        return obs, reward, done, info

    def _trade_slots_device(self):
        """`_trade_slots` without the host (device_reset): the pool -- every slot whose recipe stage is > 0 after this step's recipe launch -- is dealt out again among
        the envs that hold those slots by a random permutation, if any episode ended on this step; fixed-shape tensor ops, no readback.  Sorting the slots by
        (in the pool ? 0 : 1, index) and by (in the pool ? random key : 2 + index) lists the pool first in both orders -- in index order and in random order -- and the
        other slots identically behind it, so "the k-th env of the first order gets the k-th slot of the second" is the identity outside the pool."""
        import torch

        if not self.groups[0].pipelined:
            return
        stage = torch.cat([g.stage.reshape(-1) for g in self.groups]).to(torch.int64)
        ended = torch.cat([g.ended.reshape(-1) for g in self.groups])
        pool = stage > 0
        idx = self._arange.to(torch.float64) / float(self.B)
        by_index = torch.argsort(torch.where(pool, idx, 2.0 + idx))
        by_key = torch.argsort(torch.where(pool, torch.rand(self.B, generator=self._gen, device=self.device, dtype=torch.float64), 2.0 + idx))
        envs = self._env_of_slot_dev[by_index]
        new_slot = self._slot_dev.clone()
        new_slot[envs] = by_key
        new_slot = torch.where(ended.any(), new_slot, self._slot_dev)
        self._moved_dev += ((new_slot // self.b) != (self._slot_dev // self.b)).sum()
        inv = torch.empty_like(new_slot); inv[new_slot] = self._arange
        self._slot_dev, self._env_of_slot_dev = new_slot, inv

    @property
    def _slot_of_step(self):
        if getattr(self, "_device_trade", False):
            return self._slot_of_step_dev.cpu().numpy()
        return self._slot_of_step_host

    @_slot_of_step.setter
    def _slot_of_step(self, v):
        self._slot_of_step_host = v

    def _trade_slots(self):
        """After a step with pipelined resets: if episodes ended on it, every slot that is INSIDE the reset recipe now -- the ones that just ended and the ones that
        ended on earlier steps and are still stabilising / settling -- is dealt out again among the envs that hold those slots.  (Nothing an env owns lives in a
        slot during the recipe: its outputs are flagged `resetting` until the slot it holds at that moment reports `episode_started`, which every slot does once.)"""
        import numpy as np

        if not (self.resample and self.groups[0].pipelined) or sum(len(g.ended_rows) for g in self.groups) == 0:
            return
        pool = np.concatenate([i * self.b + np.nonzero(np.asarray(g._stage) > 0)[0] for i, g in enumerate(self.groups)])
        inv = np.empty(self.B, dtype=np.int64); inv[self._slot] = np.arange(self.B)
        self._reassign(inv[pool])

    def observe(self):
        return self._observation()

    def sync(self):
        for g in self.groups:
            g.sync()

    def status(self):
        import torch

        return self._to_envs(torch.cat([torch.maximum(g.sim.status.reshape(-1), g.solver_sim.status.reshape(-1)) for g in self.groups]))


def make_env(batch_size: int = 4096, device="cuda:0", parameters=None, constants=None, starting_seed: int = 0, apply_wrappers: bool = True, **kw):
    """`YcbRearrangeEnv.build` surface (ycb.py:96) for the batched env; accepts what envs/rearrange/blocks.py `make_env` accepts."""
    from robogym_amd.envs.rearrange.blocks import _check_supported

    parameters, constants = dict(parameters or {}), dict(constants or {})
    sp, rc = dict(parameters.get("simulation_params", {})), dict(parameters.get("robot_control_params", {}))
    if parameters.get("mesh_names") is not None or constants.get("normalize_mesh"):
        raise NotImplementedError("mesh_names / normalize_mesh: the shipped models hold fixed object sets")
    _check_supported({k: v for k, v in parameters.items() if k != "mesh_names"}, sp, rc, constants)
    args = dict(num_objects=sp.get("num_objects", 8), max_position_change=rc.get("max_position_change", 0.1), arm_reset_controller_error=rc.get("arm_reset_controller_error", True),
                n_random_initial_steps=parameters.get("n_random_initial_steps", 10), starting_seed=starting_seed, wrappers=bool(apply_wrappers),
                n_action_bins=constants.get("n_action_bins", 11), control_mode=rc.get("control_mode", "tcp+roll+yaw"), tcp_solver_mode=rc.get("tcp_solver_mode", "mocap_ik"))
    if "action_spacing" in constants:
        args["action_spacing"] = constants["action_spacing"]
    for k in ("success_threshold", "successes_needed", "success_reward", "max_timesteps_per_goal_per_obj", "use_goal_distance_reward", "goal_reward_per_object"):
        if k in constants:
            args[k] = constants[k]
    for k in ("penalty", "used_table_portion"):
        if k in sp:
            args[k] = sp[k]
    args.update(kw)
    object_sets = args.pop("object_sets", None)      # e.g. (0, 1, 2, 3): different object sets across the batch (GroupedYcbRearrangeEnv)
    if object_sets is not None and len(object_sets) > 1:
        return GroupedYcbRearrangeEnv(batch_size, device=device, object_sets=tuple(object_sets), **args)
    return BatchedYcbRearrangeEnv(batch_size, device=device, **args)


def make_simple_env(*a, **kw):
    kw["apply_wrappers"] = False
    return make_env(*a, **kw)
