# warp parity tests + profile of the warp bench (one GPU call)
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_warp_gpu.py -x -q -m gpu > gpurun_out/warp_tests.log 2>&1
tail -3 gpurun_out/warp_tests.log
timeout 300 python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline | tail -1
bash tools/gpu_prof_warp.sh
python - <<'PY'
import csv,re
for r in csv.reader(open('gpurun_out/warp_kernel_stats.csv')):
    if len(r)==5 and r[0]!='name':
        m=re.search(r'(\w+_kernel|\w+)(<|\()',r[0]); print((m.group(1) if m else r[0][:40]).ljust(36), r[1],r[2],r[3],r[4])
PY
