# rocprofv3 kernel stats of the warp bench; usage: gpu_prof_warp.sh [random|room]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
DEPTH=${1:-random}
rm -rf gpurun_out/prof_warp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_warp -o warp -- python bench.py --workload warp --warp-depth $DEPTH --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_warp.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_warp/warp_results.db gpurun_out/warp_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --workload warp --warp-depth $DEPTH --steps 20 --warmup 3 --no-cpu-baseline"
python - <<'PY'
import csv,re
for r in csv.reader(open('gpurun_out/warp_kernel_stats.csv')):
    if len(r)==5 and r[0]!='name':
        m=re.search(r'(\w+_kernel|\w+)(<|\()',r[0]); print((m.group(1) if m else r[0][:40]).ljust(36), r[1],r[2],r[3],r[4])
PY
