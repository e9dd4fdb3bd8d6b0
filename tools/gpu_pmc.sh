cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d gpurun_out/pmc1 -o pmc -- python tools/conv_bench.py > gpurun_out/pmc1.log 2>&1
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/pmc1/*.db')[0])
cur = db.cursor()
q = """select k.kernel_name as name, p.counter_name as cname, avg(p.value) as v, count(*) as n
from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id group by name, cname"""
try:
  rows = list(cur.execute(q))
except Exception as e:
  print('query failed', e)
  print([d[1] for d in cur.execute("pragma table_info(pmc_events)")])
  print([d[1] for d in cur.execute("pragma table_info(kernels)")])
  rows = []
agg = {}
for name, c, v, n in rows:
  if 'igemm' in name or 'wgrad_kernel' in name:
    short = name.split('::')[-1][:40]
    agg.setdefault(short, {})[c] = v
for k, d in agg.items():
  print(k)
  for c, v in sorted(d.items()): print('   %-28s %.3e' % (c, v))
PY
