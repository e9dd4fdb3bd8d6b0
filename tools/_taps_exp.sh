cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for D in 0 1 2 4 8 16 31; do
rm -rf gpurun_out/pw; SE3DS_COUNT_DBG=$D rocprofv3 --kernel-trace --stats -d gpurun_out/pw -o w -- python bench.py --workload warp --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
python tools/rocpd_summary.py gpurun_out/pw/w_results.db gpurun_out/pw.csv x > /dev/null 2>&1
echo "DBG=$D $(grep -E 'count_kernel|scatter_kernel|resolve_kernel' gpurun_out/pw.csv | sed -E 's/.*(splat_[a-z_]+kernel).*\)",([0-9]+),([0-9.]+),([0-9.]+),.*/\1 \4/' | tr '\n' ' ')"
done
rm -rf gpurun_out/pw gpurun_out/pw.csv
