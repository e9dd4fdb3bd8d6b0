# A/B of two builds of the library on one box under any python tool that prints its own timings:
# tools/probes/_old_lib.so against the in-tree one.   usage: lib_ab_py.sh <grep pattern> <file.py> [args]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
pat="$1"; shift
cp se3ds_amd/csrc/libse3ds_hip.so /tmp/new_lib.so
for rep in 1 2; do
  for which in old new; do
    if [ $which == old ]; then cp tools/probes/_old_lib.so se3ds_amd/csrc/libse3ds_hip.so; else cp /tmp/new_lib.so se3ds_amd/csrc/libse3ds_hip.so; fi
    echo "== $which (rep $rep)"
    timeout 600 python "$@" 2>&1 | grep "$pat" | cut -c1-200
  done
done
cp /tmp/new_lib.so se3ds_amd/csrc/libse3ds_hip.so
