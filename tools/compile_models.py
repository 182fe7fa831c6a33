"""Compile the robogym MJCF + STL assets of the hot-path envs into the flat model files shipped
under robogym_amd/models/ (the asset tree itself is not redistributed in this repository).

    ROBOGYM_ASSETS_DIR=/path/to/robogym/assets python tools/compile_models.py

Reads the assets (default: the read-only reference checkout), writes *.npz.  Derived data only:
body trees, inertias, joint/actuator/tendon tables and the convex-hull vertices of the collision
meshes in their inertial frames.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from robogym_amd.envs.dactyl.locked import MODEL_DIR, build_locked_xml  # noqa: E402
from robogym_amd.envs.dactyl.full_perpendicular import build_full_perpendicular_xml  # noqa: E402
from robogym_amd.envs.dactyl.reach import build_reach_xml  # noqa: E402
from robogym_amd.envs.rearrange.xml import YCB_SHIPPED_SETS, build_blocks_xml, build_solver_xml, load_ycb_model  # noqa: E402


def main():
    os.makedirs(MODEL_DIR, exist_ok=True)
    for name, build in (("dactyl_locked", build_locked_xml), ("dactyl_reach", build_reach_xml), ("dactyl_full_perpendicular", build_full_perpendicular_xml),
                        ("rearrange_blocks5", lambda: build_blocks_xml(5)), ("rearrange_blocks5_mocap", lambda: build_blocks_xml(5, joint_actuated=False)), ("ur16e_solver", build_solver_xml),
                        ("rearrange_ycb8", lambda: load_ycb_model(8, recompile=True)),    # (a fixed object set: envs/rearrange/xml.py YCB_MODEL_SEED)
                        ) + tuple(("rearrange_ycb8_s%d" % k, (lambda k=k: load_ycb_model(8, recompile=True, set_index=k))) for k in YCB_SHIPPED_SETS[1:]):
        model = build()
        model = model if hasattr(model, "arrays") else model.build()
        path = os.path.join(MODEL_DIR, name + ".npz")
        model.save(path)
        d = model.dims
        print("%s: nq=%d nv=%d nu=%d nbody=%d ngeom=%d nsite=%d ntendon=%d nmeshvert=%d -> %s (%d bytes)"
              % (name, d[0], d[1], d[2], d[3], d[5], d[6], d[7], d[10], path, os.path.getsize(path)))


if __name__ == "__main__":
    main()
