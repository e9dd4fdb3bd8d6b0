cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_tmp_l
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_tmp_l -o pmc -- tools/probes/build/lds_epi > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/pmc_tmp_l/*.db')[0])
rows = list(db.execute("select dispatch_id, counter_name, avg(counter_value) from pmc_events where name like '%epi_probe%' group by dispatch_id, counter_name order by dispatch_id"))
d = {}
for i, c, v in rows: d.setdefault(i, {})[c] = v
ids = sorted(d)
rbs = [128, 144, 160, 176, 192, 208, 224, 272]
for k, nm in enumerate(('park 16x16x32', 'park 32x32x16', 'write-back b128')):
  print(nm, {rb: '%.0f/%.0f' % (d[ids[k * 8 + i]].get('SQ_LDS_BANK_CONFLICT', 0) / 1e3, d[ids[k * 8 + i]].get('SQ_LDS_IDX_ACTIVE', 0) / 1e3) for i, rb in enumerate(rbs)})
PY
rm -rf gpurun_out/pmc_tmp_l
