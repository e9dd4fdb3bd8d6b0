cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
timeout 1500 python tools/step_compare.py 512 8 14 > gpurun_out/r3_g_cmp512.log 2>&1
cat gpurun_out/r3_g_cmp512.log | cut -c1-200 | grep -v "^STEP [0-9] \|^STEP 1[0-2]"
for so in 0 1; do
  SE3DS_SEGMENT_OPTIMIZER=$so timeout 600 python bench.py --no-cpu-baseline --no-warp --no-batch-max > gpurun_out/r3_g_bench_so$so.log 2>&1
  echo "segment_opt=$so: $(tail -1 gpurun_out/r3_g_bench_so$so.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["hbm_gib_peak"], d["losses"])')"
done
