#!/bin/bash
# round 5, call k: joint control + device recipe on the GPU; steady-state A/B host recipe vs device recipe
mkdir -p gpurun_out
python -m pytest tests/test_rearrange_env.py -m gpu -q -k "device_reset or device_placement or joint_control or pipelined" 2>&1 | tail -15
python tools/ab_rearrange_steady.py --steps 60 2>&1 | tail -4 | tee gpurun_out/ab_steady_blocks_r05.txt
python tools/ab_rearrange_steady.py --ycb --steps 40 2>&1 | tail -4 | tee gpurun_out/ab_steady_ycb_r05.txt
