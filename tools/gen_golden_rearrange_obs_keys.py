"""tests/golden/rearrange_obs_keys.json: the key order of `RearrangeEnv._observe_simple` (/root/reference/robogym/envs/rearrange/common/base.py:376-421) and, per key,
which simulation / robot accessor feeds it — obtained by executing the method's own source on a recording stub.  Needs /root/reference; the fixture travels.

    python tools/gen_golden_rearrange_obs_keys.py
"""
import ast
import json
import os
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BASE = "/root/reference/robogym/envs/rearrange/common/base.py"


class Tag(np.ndarray):
    def __array_finalize__(self, obj):
        self.tag = getattr(obj, "tag", None)


def tagged(name):
    a = np.zeros(1).view(Tag)
    a.tag = name
    return a


class Recorder:
    def __init__(self, prefix):
        self._p = prefix

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self._p + "." + name
        if name == "robot":
            return Recorder(full)
        if name == "qpos":
            return tagged(full)

        def call(*a, **k):
            return Recorder(full + "()") if name == "observe" else tagged(full + "()")
        return call


class GoalDict(dict):
    def __getitem__(self, k):
        return tagged("goal[%s]" % k)


def main():
    tree = ast.parse(open(BASE).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "RearrangeEnv"][0]
    meth = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "_observe_simple"]
    m = ast.Module(body=[ast.ClassDef(name="RearrangeEnv", bases=[], keywords=[], body=meth, decorator_list=[])], type_ignores=[])
    ast.fix_missing_locations(m)
    ns = {"np": np}
    exec(compile(m, BASE, "exec"), ns)
    env = types.SimpleNamespace(mujoco_simulation=Recorder("sim"), _goal=GoalDict(), _goal_info_dict=GoalDict(), _is_goal_achieved=False,
                                constants=types.SimpleNamespace(mask_obs_outside_placement_area=False))
    obs = ns["RearrangeEnv"]._observe_simple(env)
    out = [[k, (getattr(v, "tag", None) or ("np.array([...], %s)" % v.dtype))] for k, v in obs.items()]
    for i, (k, src) in enumerate(out):
        if src.startswith("np.array"):
            out[i][1] = {"is_goal_achieved": "self._is_goal_achieved (int32)", "safety_stop": "robot.observe().is_in_safety_stop()"}[k]
    path = os.path.join(HERE, "..", "tests", "golden", "rearrange_obs_keys.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, len(out), "keys")


if __name__ == "__main__":
    main()
