# A/B of the batched epilogue write-back (round 5) inside the serial instrumented step, one box:
# the -DSE3DS_PROBE build of conv.hip carries both forms (SE3DS_PROBE_MODE=4: the serial loop); loaded
# through SE3DS_LIB, the in-tree library is never touched.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# (the probe object is built here: the box has hipcc, and an untracked prebuilt object went stale)
mkdir -p /tmp/probe_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -DSE3DS_PROBE -Iinclude -c se3ds_amd/csrc/conv.hip -o /tmp/probe_obj/conv_probe.o || exit 1
objs=$(ls se3ds_amd/csrc/_obj/*.o | grep -v conv.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libprobe.so /tmp/probe_obj/conv_probe.o $objs || exit 1
export SE3DS_LIB=/tmp/libprobe.so
bash tools/probes/shapes_ab.sh ${1:-epi} SE3DS_PROBE_MODE=0 SE3DS_PROBE_MODE=4
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for m in 0 4; do echo "== default schedule, SE3DS_PROBE_MODE=$m"; SE3DS_PROBE_MODE=$m timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-batch-max --no-warp --no-shipped --no-fp32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms/step %.2f' % d['ms_per_step'])"; done; done
