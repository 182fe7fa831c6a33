#!/bin/bash
# Round 5: sections of the Hessian assembly on the large configuration (-DRB_HESS_PROBE=1 static rows | 2 contact weights | 3 the contacts' entries)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for k in 1 2 3; do
  echo "probe $k"; RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_probe$k.so python tools/large_stage_profile.py 512 2>&1 | grep -E "inside Newton|means" | cut -c1-330 | tee gpurun_out/large_hess_probe$k.txt
done
