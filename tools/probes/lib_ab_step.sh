# A/B of two builds of the library on one box, whole step: tools/probes/_old_lib.so against the in-tree
# one (2 reps each, interleaved).   usage: lib_ab_step.sh [extra bench.py args]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp se3ds_amd/csrc/libse3ds_hip.so /tmp/new_lib.so
for rep in 1 2; do
  for which in old new; do
    if [ $which == old ]; then cp tools/probes/_old_lib.so se3ds_amd/csrc/libse3ds_hip.so; else cp /tmp/new_lib.so se3ds_amd/csrc/libse3ds_hip.so; fi
    echo "== $which (rep $rep)"
    timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-batch-max --no-warp --no-shipped --no-fp32 "$@" 2>gpurun_out/ab.err | python -c "import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('ms/step %.2f value %.3f frac %.4f conv_ms %.2f in_step %.4f' % (d['ms_per_step'], d['value'], r['frac'], r['conv_ms_per_step'], r['frac_in_step']))" || tail -5 gpurun_out/ab.err
  done
done
cp /tmp/new_lib.so se3ds_amd/csrc/libse3ds_hip.so
