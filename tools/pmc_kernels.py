"""Per-kernel averages of one PMC counter from rocprofv3's rocpd sqlite, MB per launch (the
counters report KB).  usage: pmc_kernels.py '<glob of .db>' '<regex on kernel names>'"""
import glob
import re
import sqlite3
import sys

pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for path in glob.glob(sys.argv[1]):
  db = sqlite3.connect(path)
  q = 'select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name'
  for name, counter, v, n in db.execute(q):
    if pat and not pat.search(name):
      continue
    m = re.search(r'(\w+_kernel)', name)
    print((m.group(1) if m else name[:40]).ljust(30), counter, '%.1f MB (n=%d)' % (v / 1024, n))
