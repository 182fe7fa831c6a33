"""A/B of the rearrange steady-state lines (bench.bench_rearrange_steady): host recipe (one readback of the flags per step) against the device recipe
(ra_recipe_kernel, no readback).  Usage: python tools/ab_rearrange_steady.py [--ycb] [--steps 60]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ycb", action="store_true")
ap.add_argument("--steps", type=int, default=60)
a = ap.parse_args()


class A:
    batch, steps = 4096, a.steps


for dr in (False, True, False, True):
    out = bench.bench_rearrange_steady(A, ycb=a.ycb, device_reset=dr)
    print(json.dumps({"device_reset": dr, "value": out["value"], "ms_per_step": out["ms_per_step"], **{k: out["config"][k] for k in (
        "fraction_of_env_steps_inside_the_reset_recipe", "episodes_ended_in_window", "episodes_started_in_window", "status_bits", "placements_out_of_trials")}}))
