# per-phase timing of the conv kernels with a -DSE3DS_PROBE build of conv.hip (built here:
#   hipcc ... -DSE3DS_PROBE -c se3ds_amd/csrc/conv.hip -o tools/probes/build/conv_probe.o)
# linked with the other objects of the in-tree build into a scratch library that replaces the
# in-tree one for the duration of the run.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
objs=$(ls se3ds_amd/csrc/_obj/*.o | grep -v conv.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libprobe.so tools/probes/build/conv_probe.o $objs || exit 1
cp se3ds_amd/csrc/libse3ds_hip.so /tmp/lib_keep.so
cp /tmp/libprobe.so se3ds_amd/csrc/libse3ds_hip.so
timeout 900 python tools/conv_phases.py
cp /tmp/lib_keep.so se3ds_amd/csrc/libse3ds_hip.so
