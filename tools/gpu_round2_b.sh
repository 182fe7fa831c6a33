cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -q -s 2>&1 | tail -150) > gpurun_out/r02_gputest2.log 2>&1
python tools/ncon_hist.py > gpurun_out/r02_ncon_hist.txt 2>&1
tail -5 gpurun_out/r02_gputest2.log
