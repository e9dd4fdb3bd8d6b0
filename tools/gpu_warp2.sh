cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_warp_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -3
python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('warp value %.1f p/s, project+splat %.1f us, %.0f GB/s (%.1f%% HBM)' % (d['value'], r['ms_per_launch']*1e3, r['achieved'], 100*r['frac']))"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_warp2 -o w -- python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python tools/rocpd_summary.py gpurun_out/prof_warp2/w_results.db gpurun_out/warp_v1_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline"
cut -c1-70,150-400 gpurun_out/warp_v1_kernel_stats.csv | head -12
