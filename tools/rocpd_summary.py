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

# footer: the convolution family as one line, to compare with bench.py's roofline block
# (bench times API calls on the launch stream: a wgrad call = its kernel + the split reduce)
conv = [r for r in rows if any(t in r[0] for t in ('igemm', 'wgrad'))]
if conv:
  calls = sum(r[1] for r in conv)
  tot = sum(r[2] for r in conv)
  main = sum(r[1] for r in conv if 'reduce' not in r[0] and 'fixup' not in r[0])
  out.write('# conv family: %d kernel launches (%d without reduce/fix-up), %.3f us total, %.3f us per main launch\n'
            % (calls, main, tot, tot / max(main, 1)))
