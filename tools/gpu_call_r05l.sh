#!/bin/bash
# Round 5, call l: A/B of the substep-granular dispatch's row stores -- plain stores + vmcnt drain (default) against agent-scope (sc1) atomic stores (-DRG_ROW_ST_SC1),
# alternating on one box; then the dispatch identity tests on the sc1 build
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/ab_items_sc1_r05.txt
for rep in 1 2 3; do
  for v in default sc1; do
    L=$GRAFT_REPO_ROOT/robogym_amd/csrc/librgstep.so; [ $v = sc1 ] && L=$GRAFT_REPO_ROOT/ab_libs/librgstep_sc1.so
    RGSTEP_LIB=$L python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v rep $rep: %.0f env-steps/s, %.3f ms per step, kernel %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" | tee -a gpurun_out/ab_items_sc1_r05.txt
  done
done
RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_sc1.so python -m pytest tests/test_kernel_emul.py tests/test_gpu_parity.py -m gpu -q -k "items or unserved or resync or distinct" 2>&1 | tail -3 | tee -a gpurun_out/ab_items_sc1_r05.txt
