cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_tmp_l
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT -d gpurun_out/pmc_tmp_l -o pmc -- tools/probes/build/lds_swz > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/pmc_tmp_l/*.db')[0])
rows = list(db.execute("select dispatch_id, avg(counter_value) from pmc_events where name like '%swz_probe%' group by dispatch_id order by dispatch_id"))
vals = [v for _, v in rows]
print(len(vals), 'dispatches')
for cand in range(len(vals) // 16):
  seg = vals[cand * 16:(cand + 1) * 16]
  print('swizzle %d: start rows with conflicts: %s' % (cand, [r for r, v in enumerate(seg) if v > 0]))
PY
rm -rf gpurun_out/pmc_tmp_l
