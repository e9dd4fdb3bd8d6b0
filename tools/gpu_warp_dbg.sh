cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for d in 0 1 2 3; do
SE3DS_SPLAT_DEBUG=$d rocprofv3 --kernel-trace --stats -d gpurun_out/wdbg$d -o w -- python bench.py --workload warp --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python tools/rocpd_summary.py gpurun_out/wdbg$d/w_results.db /dev/stdout | grep -E "zmin|resolve" | cut -c1-60,200-400 | sed "s/^/dbg=$d /"
done
