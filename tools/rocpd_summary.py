"""Dump the kernel summary of a rocprofv3 rocpd .db as CSV (name, calls, total_us, avg_us, pct)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
if len(sys.argv) > 3:
  out.write('# ' + sys.argv[3] + '\n')
out.write('name,calls,total_us,avg_us,pct\n')
for r in rows:
  out.write('"%s",%d,%.3f,%.3f,%.2f\n' % (r[0].replace('"', "'"), r[1], r[2], r[3], r[4]))
