cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_nets_gpu.py -m gpu -q -x --tb=line -k "conv or norm or train_g_d" 2>&1 | tail -3 | cut -c1-300
for b in ${BATCHES:-4}; do
timeout 900 python bench.py --steps 2 --warmup 1 --batch $b --no-cpu-baseline > gpurun_out/bench_gan_b$b.log 2>&1
tail -1 gpurun_out/bench_gan_b$b.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('batch', d['config']['per_gpu_batch'], 'value %.2f p/s  ms/step %.1f  conv_ms %.1f  conv TF/s %.1f  hbm %.1f GiB' % (d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['achieved'], d['hbm_gib_peak']))
for k,v in r['by_kind'].items(): print('   ', k, 'ms %.1f tflops %.1f n %d' % (v['ms'], v['tflops'], v['launches']))
" || tail -5 gpurun_out/bench_gan_b$b.log
done
