# A/B of two builds of the library on one box, whole step: tools/probes/_old_lib.so (untracked: copy a
# previous build there) against the in-tree one, 2 reps each, interleaved.  The other build is selected
# with SE3DS_LIB -- the in-tree file is never touched.   usage: lib_ab_step.sh [extra bench.py args]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
[ -f tools/probes/_old_lib.so ] || { echo "tools/probes/_old_lib.so missing"; exit 1; }
for rep in 1 2; do
  for which in old new; do
    if [ $which == old ]; then export SE3DS_LIB=$GRAFT_REPO_ROOT/tools/probes/_old_lib.so; else unset SE3DS_LIB; fi
    echo "== $which (rep $rep)"
    timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-batch-max --no-warp --no-shipped --no-fp32 "$@" 2>gpurun_out/ab.err | python -c "import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('ms/step %.2f value %.3f frac %.4f conv_ms %.2f in_step %.4f' % (d['ms_per_step'], d['value'], r['frac'], r['conv_ms_per_step'], r['frac_in_step']))" || tail -5 gpurun_out/ab.err
  done
done
unset SE3DS_LIB
