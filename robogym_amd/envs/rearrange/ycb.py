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
    return BatchedYcbRearrangeEnv(batch_size, device=device, **args)


def make_simple_env(*a, **kw):
    kw["apply_wrappers"] = False
    return make_env(*a, **kw)
