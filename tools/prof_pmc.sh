cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oiE "SQC?_[A-Z_0-9]*(ICACHE|IFETCH|INST_CACHE|WAIT_INST|WAIT_ANY|ACTIVE_INST|INSTS_VALU|INSTS_SALU|INSTS_LDS|INSTS_SMEM|INSTS_VMEM|WAVE_CYCLES|BUSY_CYCLES|INST_LEVEL)[A-Z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/counters.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d gpurun_out/pmc1 -o pmc1 --output-format csv -- python tools/stage_profile.py 8192 > gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES -d gpurun_out/pmc2 -o pmc2 --output-format csv -- python tools/stage_profile.py 8192 > gpurun_out/pmc2.log 2>&1
ls -R gpurun_out | head -30
