"""Per-kernel averages of every PMC counter in a rocprofv3 rocpd .db (conv kernels only by default)."""
import sqlite3, sys, glob
db = sqlite3.connect(glob.glob(sys.argv[1])[0])
pat = sys.argv[2] if len(sys.argv) > 2 else 'igemm|wgrad'
cur = db.cursor()
cols = [d[1] for d in cur.execute("pragma table_info(pmc_events)")]
rows = list(cur.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name"))
agg = {}
for name, c, v, n in rows:
  if any(t in name for t in pat.split('|')):
    short = name.split('::')[-1][:48]
    agg.setdefault(short, {})[c] = (v, n)
for k, d in agg.items():
  print(k)
  for c, (v, n) in sorted(d.items()): print('   %-28s %.4e  (n=%d)' % (c, v, n))
