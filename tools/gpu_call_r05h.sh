#!/bin/bash
# Round 5, eighth GPU call: where the Hessian assembly's cycles go (-DRB_HESS_PROBE=1 static rows | 2 contact weights | 3 the contacts' entries)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for k in 1 2 3; do
  RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_probe$k.so python tools/rearrange_stage_profile.py 4096 2>&1 | grep -E "world|inside Newton" | cut -c1-330 > gpurun_out/hess_probe$k.txt
  echo "probe $k"; cat gpurun_out/hess_probe$k.txt
done
