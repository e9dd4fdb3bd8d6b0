# per-phase timing of the conv kernels with a -DSE3DS_PROBE build of conv.hip, linked with the other
# objects of the in-tree build into a scratch library that THIS run loads through SE3DS_LIB (the
# in-tree library is never touched).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# (the probe object is built here: the box has hipcc, and an untracked prebuilt object went stale)
mkdir -p /tmp/probe_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -DSE3DS_PROBE -Iinclude -c se3ds_amd/csrc/conv.hip -o /tmp/probe_obj/conv_probe.o || exit 1
objs=$(ls se3ds_amd/csrc/_obj/*.o | grep -v conv.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libprobe.so /tmp/probe_obj/conv_probe.o $objs || exit 1
export SE3DS_LIB=/tmp/libprobe.so
timeout 900 python tools/conv_phases.py
