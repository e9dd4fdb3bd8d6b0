# A/B of two builds of the library on one box under any python tool that prints its own timings:
# tools/probes/_old_lib.so (untracked) against the in-tree one, selected with SE3DS_LIB.
# usage: lib_ab_py.sh <grep pattern> <file.py> [args]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
[ -f tools/probes/_old_lib.so ] || { echo "tools/probes/_old_lib.so missing"; exit 1; }
pat="$1"; shift
for rep in 1 2; do
  for which in old new; do
    if [ $which == old ]; then export SE3DS_LIB=$GRAFT_REPO_ROOT/tools/probes/_old_lib.so; else unset SE3DS_LIB; fi
    echo "== $which (rep $rep)"
    timeout 600 python "$@" 2>&1 | grep "$pat" | cut -c1-200
  done
done
unset SE3DS_LIB
