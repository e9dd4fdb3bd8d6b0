cd $GRAFT_REPO_ROOT
for i in 1 2; do
python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('batch', d['config']['per_gpu_batch'], 'value %.2f p/s  ms/step %.1f  conv_ms %.1f  conv TF/s %.1f' % (d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['achieved']))"
done
