"""MFMA utilisation and shader clock of every kernel INSIDE the training step.
usage: pmc_step.py <rocpd .db glob> out.json [top_n]
The .db comes from ONE pass of
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python bench.py ...
(counter collection serialises the dispatches: every kernel runs alone, but on the step's own
tensors, cache state and -- what the isolated tools/one_conv.py loops cannot show -- under the
step's sustained power draw).  Per kernel name: launches, average duration from the kernel trace,
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES x 32 / (1024 SIMDs x SQ_BUSY_CYCLES)   (units: profiles/r02_conv_pmc_summary.md)
  clock_ghz = SQ_BUSY_CYCLES / duration
"""
import glob
import json
import re
import sqlite3
import sys

path = glob.glob(sys.argv[1])[0]
out_path = sys.argv[2]
top_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
db = sqlite3.connect(path)
dur = {}
for name, calls, total, avg in db.execute('select name,total_calls,total_duration,average from top_kernels'):
  dur[name] = dict(calls=calls, total_us=total, avg_us=avg)   # (rocpd_summary.py: these columns are in us)
pmc = {}
for name, counter, v, n in db.execute(
    'select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name'):
  pmc.setdefault(name, {})[counter] = v
rows = []
for name, d in dur.items():
  c = pmc.get(name, {})
  busy, mf = c.get('SQ_BUSY_CYCLES'), c.get('SQ_VALU_MFMA_BUSY_CYCLES')
  m = re.search(r'(\w+_kernel(<[^(]*>)?)', name)
  short = (m.group(1) if m else name[:70]).replace('se3ds::(anonymous namespace)::', '')
  r = dict(kernel=short, calls=d['calls'], avg_us=d['avg_us'], total_ms=d['total_us'] / 1e3)
  if busy:
    r['clock_ghz'] = busy / (d['avg_us'] * 1e3)
    if mf is not None:
      r['mfma_util'] = mf * 32.0 / (1024.0 * busy)
  rows.append(r)
rows.sort(key=lambda r: -r['total_ms'])
tot = sum(r['total_ms'] for r in rows)
conv = [r for r in rows if 'mfma_util' in r and r['mfma_util'] > 0.01]
wsum = sum(r['total_ms'] for r in conv)
summary = dict(
    total_kernel_ms=tot,
    mfma_kernels_ms=wsum,
    mfma_util_time_weighted=sum(r['mfma_util'] * r['total_ms'] for r in conv) / max(wsum, 1e-9),
    clock_ghz_time_weighted_mfma_kernels=sum(r.get('clock_ghz', 0) * r['total_ms'] for r in conv) / max(wsum, 1e-9))
json.dump(dict(summary=summary, kernels=rows[:top_n]), open(out_path, 'w'), indent=1)
print(json.dumps(summary))
for r in rows[:top_n]:
  print('%-64s %6d x %8.1f us = %8.1f ms  clk %s  mfma %s' % (
      r['kernel'][:64], r['calls'], r['avg_us'], r['total_ms'],
      ('%.2f' % r['clock_ghz']) if 'clock_ghz' in r else '  - ',
      ('%.3f' % r['mfma_util']) if 'mfma_util' in r else '  -  '))
