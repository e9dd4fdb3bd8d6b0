cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 4 8; do
timeout 900 python bench.py --steps 2 --warmup 1 --batch $b --no-cpu-baseline > gpurun_out/bench_gan_b$b.log 2>&1
tail -1 gpurun_out/bench_gan_b$b.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('batch', d['config']['per_gpu_batch'], 'value %.2f p/s  ms/step %.1f  conv_ms %.1f  conv TF/s %.1f  hbm %.1f GiB' % (d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['achieved'], d['hbm_gib_peak']))
for k,v in r['by_kind'].items(): print('   ', k, 'ms %.1f tflops %.1f n %d' % (v['ms'], v['tflops'], v['launches']))
"
done
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_gan -o gan -- python bench.py --steps 2 --warmup 1 --batch 4 --no-cpu-baseline > gpurun_out/prof_gan.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_gan/gan_results.db gpurun_out/gan_b4_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --batch 4 --no-cpu-baseline (4 train_g_d steps incl. warmup+profiled step)"
head -30 gpurun_out/gan_b4_kernel_stats.csv | cut -c1-200
