"""Per-STEP kernel statistics from a rocprofv3 --kernel-trace rocpd .db of bench.py (gan_step).

  python tools/step_trace.py <results.db> <out.csv> [first_step last_step] ["note"]

`rocprofv3 --stats` sums over the whole process: model build (thousands of torch fill / copy /
random kernels of the parameter initialisers), warm-up and the instrumented steps included, so
"launches per step = calls / steps" over-counts whatever the build launches.  This tool cuts the
trace at a kernel that runs exactly once per generator forward pass (the encoder's max-pool) and
averages the windows [marker k, marker k+1) for k in first_step .. last_step (default 4 .. 11: timed
steps of the default bench: 3 warm-up + 10 timed + 2 instrumented).  Output per kernel: launches per
step, ms per step, average us; footer: launches per step, sum of the kernels shorter than 15 us,
the norm_* family, the convolution family, torch / runtime kernels."""
import collections
import re
import sqlite3
import sys

MARKER = 'maxpool_fwd_kernel<unsigned short>'


def short(name):
  s = re.sub(r'\(anonymous namespace\)::', '', name)
  s = re.sub(r'^void ', '', s)
  s = re.sub(r'se3ds::', '', s)
  cut = s.find('(')
  # keep template arguments, drop the parameter list
  depth, out = 0, []
  for ch in s:
    if ch == '<':
      depth += 1
    elif ch == '>':
      depth -= 1
    elif ch == '(' and depth == 0:
      break
    out.append(ch)
  s = ''.join(out) if cut >= 0 else s
  return s[:110]


def family(n):
  if n.startswith('norm_'):
    return 'norm'
  if any(t in n for t in ('igemm', 'wgrad', 'thin_', 'weight_prep')):
    return 'conv'
  if any(t in n for t in ('adam', 'clip', 'sn_', 'sqsum', 'ema_')):
    return 'optimiser'
  if 'at::' in n or 'rocclr' in n or 'hip' in n.lower()[:6]:
    return 'torch/runtime'
  return 'other'


def main():
  db = sqlite3.connect(sys.argv[1])
  out = open(sys.argv[2], 'w') if len(sys.argv) > 2 and sys.argv[2] != '-' else sys.stdout
  k0 = int(sys.argv[3]) if len(sys.argv) > 4 else 4
  k1 = int(sys.argv[4]) if len(sys.argv) > 4 else 11
  note = sys.argv[5] if len(sys.argv) > 5 else ''
  views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
  src = 'kernels' if 'kernels' in views else next(v for v in views if 'kernel_dispatch' in v)
  cols = [r[1] for r in db.execute(f'pragma table_info({src})')]
  name_col = 'name' if 'name' in cols else next(c for c in cols if 'name' in c)
  rows = list(db.execute(f'select start, end, {name_col} from {src} order by start'))
  marks = [r[0] for r in rows if MARKER in r[2]]
  if len(marks) < k1 + 2:
    k1 = len(marks) - 2
    k0 = min(k0, k1)
  lo, hi = marks[k0], marks[k1 + 1]
  nsteps = k1 + 1 - k0
  agg = collections.OrderedDict()
  first = last = None
  for s, e, n in rows:
    if s < lo or s >= hi:
      continue
    a = agg.setdefault(short(n), [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    first = s if first is None else first
    last = max(last or e, e)
  items = sorted(agg.items(), key=lambda kv: -kv[1][1])
  if note:
    out.write('# ' + note + '\n')
  out.write(f'# steady-state windows {k0}..{k1} ({nsteps} steps) cut at {MARKER}; wall per step '
            f'{(hi - lo) / 1e6 / nsteps:.2f} ms\n')
  out.write('name,launches_per_step,ms_per_step,avg_us\n')
  tot_l = tot_ms = small_l = small_ms = 0.0
  fam = collections.OrderedDict()
  for n, (cnt, us) in items:
    lps, ms, avg = cnt / nsteps, us / 1e3 / nsteps, us / cnt
    out.write('"%s",%.1f,%.3f,%.2f\n' % (n.replace('"', "'"), lps, ms, avg))
    tot_l += lps
    tot_ms += ms
    if avg < 15.0:
      small_l += lps
      small_ms += ms
    f = fam.setdefault(family(n), [0.0, 0.0])
    f[0] += lps
    f[1] += ms
  out.write('# launches per step %.0f, sum of kernel time %.2f ms per step\n' % (tot_l, tot_ms))
  out.write('# kernels with an average below 15 us: %.0f launches, %.2f ms per step\n' % (small_l, small_ms))
  for f, (l, ms) in fam.items():
    out.write('# family %-14s %6.0f launches %8.2f ms per step\n' % (f, l, ms))


if __name__ == '__main__':
  main()
