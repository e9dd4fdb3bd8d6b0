# round 3: where the 128-channel halo kernel spends its time -- variants with the global stores,
# the whole epilogue or the MFMAs compiled out (timing only, results are garbage)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
cp se3ds_amd/csrc/libse3ds_hip.so /tmp/base.so
for v in base nostore noepi nomfma2 base; do
  if [ $v = base ]; then cp /tmp/base.so se3ds_amd/csrc/libse3ds_hip.so; else cp se3ds_amd/csrc/_exp/lib_$v.so se3ds_amd/csrc/libse3ds_hip.so; fi
  echo "== $v"
  N=8 python tools/conv_bench.py 2>&1 | grep "128->128 @\|1024->1024"
done
cp /tmp/base.so se3ds_amd/csrc/libse3ds_hip.so
