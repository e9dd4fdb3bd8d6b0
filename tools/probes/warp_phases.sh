# per-phase timing of the two sorted-splat kernels: a -DSE3DS_PROBE build of geom.hip (built in the
# container: tools/probes/build/geom_probe.o, ~1 min; rebuilt here when missing) linked with the other
# objects of the in-tree build into a scratch library that THIS run loads through SE3DS_LIB.
#   gpurun -- 'bash tools/probes/warp_phases.sh [height ...]'
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
obj=tools/probes/build/geom_probe.o
if [ ! -f $obj ] || [ se3ds_amd/csrc/geom.hip -nt $obj ]; then
  mkdir -p tools/probes/build
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -DSE3DS_PROBE -c se3ds_amd/csrc/geom.hip -o $obj || exit 1
fi
objs=$(ls se3ds_amd/csrc/_obj/*.o | grep -v geom.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libgeomprobe.so $obj $objs || exit 1
export SE3DS_LIB=/tmp/libgeomprobe.so
for hh in ${@:-512 1024}; do
  for d in random room; do timeout 300 python tools/warp_phases.py --height $hh --depth $d; done
done
