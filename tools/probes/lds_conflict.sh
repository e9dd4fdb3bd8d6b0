cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_tmp_l
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES -d gpurun_out/pmc_tmp_l -o pmc -- tools/probes/build/lds_conflict > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob, re
db = sqlite3.connect(glob.glob('gpurun_out/pmc_tmp_l/*.db')[0])
rows = list(db.execute("select name, counter_name, avg(counter_value) from pmc_events group by name, counter_name"))
agg = {}
for n, c, v in rows:
  m = re.search(r'probe<(\d+)>', n)
  if m: agg.setdefault(int(m.group(1)), {})[c] = v
for k in sorted(agg):
  d = agg[k]
  print('pattern %d: conflict %.3g  idx_active %.3g  insts %.3g  busy %.3g   conflict/inst %.2f' % (
      k, d.get('SQ_LDS_BANK_CONFLICT', 0), d.get('SQ_LDS_IDX_ACTIVE', 0), d.get('SQ_INSTS_LDS', 0), d.get('SQ_BUSY_CYCLES', 0),
      d.get('SQ_LDS_BANK_CONFLICT', 0) / max(d.get('SQ_INSTS_LDS', 1), 1)))
PY
rm -rf gpurun_out/pmc_tmp_l
