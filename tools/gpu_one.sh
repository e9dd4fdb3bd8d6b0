# kernel-level timing of one conv shape: gpu_one.sh cin cout k s h w pad n
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
rm -rf gpurun_out/prof_one
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_one -o one -- python tools/one_conv.py "$@" > gpurun_out/prof_one.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_one/one_results.db gpurun_out/one_kernel_stats.csv "one_conv $*"
python - <<'PY'
import csv,re
for r in list(csv.reader(open('gpurun_out/one_kernel_stats.csv')))[:14]:
    if len(r)==5 and r[0]!='name':
        n=re.sub(r'\(.*','',re.sub(r'se3ds::\(anonymous namespace\)::','',r[0]))[:60]
        print(n.ljust(60), r[1],r[2],r[3],r[4])
PY
