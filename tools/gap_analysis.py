"""Device idle time between kernels, from a rocprofv3 --kernel-trace rocpd .db.

  python tools/gap_analysis.py <results.db> [skip_frac]

Takes the kernel dispatches in the last (1 - skip_frac) of the trace's time range (default 0.5:
the steady-state steps), merges their [start, end) intervals over all streams / queues and prints
the busy time, the idle time, a histogram of the idle gaps and the kernels that most often END
right before a long gap.  Answers "is the step GPU-bound or are there launch bubbles a hipGraph
would remove" for DESIGN.md section 3.4."""
import collections
import sqlite3
import sys


def main():
  db = sqlite3.connect(sys.argv[1])
  skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
  views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
  src = 'kernels' if 'kernels' in views else next(v for v in views if 'kernel_dispatch' in v)
  cols = [r[1] for r in db.execute(f'pragma table_info({src})')]
  if 'start' not in cols or 'end' not in cols:
    print('columns of', src, cols)
    return
  name_col = 'name' if 'name' in cols else next(c for c in cols if 'name' in c)
  extra = next((c for c in ('stream_id', 'queue_id', 'stream') if c in cols), None)
  q = f'select start, end, {name_col}' + (f', {extra}' if extra else '') + f' from {src} order by start'
  rows = list(db.execute(q))
  t0, t1 = rows[0][0], max(r[1] for r in rows)
  cut = t0 + skip * (t1 - t0)
  rows = [r for r in rows if r[0] >= cut]
  span = max(r[1] for r in rows) - rows[0][0]
  busy, cur_s, cur_e, last_name = 0, rows[0][0], rows[0][1], rows[0][2]
  gaps = []
  for r in rows[1:]:
    if r[0] > cur_e:
      busy += cur_e - cur_s
      gaps.append((r[0] - cur_e, last_name, r[2]))
      cur_s, cur_e, last_name = r[0], r[1], r[2]
    elif r[1] > cur_e:
      cur_e, last_name = r[1], r[2]
  busy += cur_e - cur_s
  ksum = sum(r[1] - r[0] for r in rows)
  print(f'kernels {len(rows)}  span {span / 1e6:.2f} ms  busy (union) {busy / 1e6:.2f} ms  '
        f'idle {(span - busy) / 1e6:.2f} ms ({100 * (span - busy) / span:.1f} %)  '
        f'sum of durations {ksum / 1e6:.2f} ms (overlap {(ksum - busy) / 1e6:.2f} ms)')
  if extra:
    per = collections.Counter()
    for r in rows:
      per[r[3]] += r[1] - r[0]
    print('per', extra, {k: round(v / 1e6, 2) for k, v in per.items()})
  edges = [1, 2, 4, 8, 16, 32, 64, 128, 1 << 30]
  hist = collections.Counter()
  tot = collections.Counter()
  for g, _, _ in gaps:
    us = g / 1e3
    b = next(e for e in edges if us < e)
    hist[b] += 1
    tot[b] += us
  lo = 0
  for e in edges:
    print(f'  gaps {lo:4d}-{e if e < 1 << 30 else "inf":>4} us: {hist[e]:6d}  total {tot[e] / 1e3:8.2f} ms')
    lo = e
  before = collections.Counter()
  for g, a, b in gaps:
    if g > 8e3:
      before[(a[:60], b[:60])] += g / 1e6
  print('largest contributors (kernel before gap -> kernel after), ms:')
  for (a, b), v in before.most_common(12):
    print(f'  {v:7.2f}  {a} -> {b}')


if __name__ == '__main__':
  main()
