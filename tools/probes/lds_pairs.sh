cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_tmp_l
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT -d gpurun_out/pmc_tmp_l -o pmc -- tools/probes/build/lds_pairs > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/pmc_tmp_l/*.db')[0])
cols = [d[1] for d in db.execute("pragma table_info(pmc_events)")]
print(cols)
key = 'dispatch_id' if 'dispatch_id' in cols else ('event_id' if 'event_id' in cols else cols[0])
rows = list(db.execute(f"select {key}, avg(counter_value) from pmc_events where name like '%pair_probe%' group by {key} order by {key}"))
vals = [v for _, v in rows]
print(len(vals))
for a0, seg in ((0, vals[:64]), (16, vals[64:128])):
  print('lanes conflicting with lane', a0, ':', [b for b, v in enumerate(seg) if v > 0])
PY
rm -rf gpurun_out/pmc_tmp_l
