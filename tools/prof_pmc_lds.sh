cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oiE "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/lds_counters.txt
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL -d gpurun_out/pmc3 -o pmc3 --output-format csv -- python tools/stage_profile.py 8192 > gpurun_out/pmc3.log 2>&1
