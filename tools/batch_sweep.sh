# env-steps/s of dactyl/locked over the batch size (one MI355X): bash tools/batch_sweep.sh > gpurun_out/batch_sweep.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
echo "B  env-steps/s  ms/step  kernel ms  (bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch B; 256 CUs x 11 resident envs = 2816 per pass)"
for B in 256 1024 2048 2816 4096 5632 8192 8448 11264 16384 32768; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch $B 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('%6d  %9.0f  %7.3f  %7.3f   passes %.2f' % ($B, r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], $B / 2816.0))"
done
