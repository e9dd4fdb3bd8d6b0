"""Summarise rocprofv3 PMC passes (rocpd .db files) per kernel: counter averages per launch.
usage: pmc_summary.py out.json [--source=se3ds_amd/csrc/x.hip] '<glob>' ['<regex on kernel names>']
(several globs allowed; --source records the kernel file's sha1 under "_meta" so that bench.py can
tell whether the summary still describes the kernels it runs)
FETCH_SIZE / WRITE_SIZE are reported in MB per launch (the counters are in KB); FETCH_SIZE is also
given with the guide's gfx950 correction for wide coalesced reads (x2, MI355X_MICROARCH.md, HBM)."""
import glob
import json
import re
import sqlite3
import sys

import hashlib
import os

out_path = sys.argv[1]
argv = [a for a in sys.argv[2:] if not a.startswith('--source=')]
source = next((a.split('=', 1)[1] for a in sys.argv[2:] if a.startswith('--source=')), None)
globs = [a for a in argv if '*' in a or a.endswith('.db')]
pats = [a for a in argv if a not in globs]
pat = re.compile(pats[0]) if pats else None
res = {}
for g in globs:
  for path in glob.glob(g):
    db = sqlite3.connect(path)
    q = 'select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name'
    for name, counter, v, n in db.execute(q):
      if pat and not pat.search(name):
        continue
      m = re.search(r'(\w+_kernel(<[^(]*>)?)', name)
      short = (m.group(1) if m else name[:60]).replace('se3ds::(anonymous namespace)::', '')
      res.setdefault(short, {})[counter] = dict(avg=v, launches=n)
for k, d in res.items():
  if 'FETCH_SIZE' in d:
    d['fetch_mb'] = d['FETCH_SIZE']['avg'] / 1024
    d['fetch_mb_x2_wide_read_correction'] = 2 * d['fetch_mb']
  if 'WRITE_SIZE' in d:
    d['write_mb'] = d['WRITE_SIZE']['avg'] / 1024
  if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'SQ_BUSY_CYCLES' in d and d['SQ_BUSY_CYCLES']['avg'] > 0:
    d['mfma_busy_over_sq_busy'] = d['SQ_VALU_MFMA_BUSY_CYCLES']['avg'] / d['SQ_BUSY_CYCLES']['avg']
  if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d and d['GRBM_GUI_ACTIVE']['avg'] > 0:
    # MFMA_BUSY sums over the 4 SIMDs of 256 CUs; GUI_ACTIVE is wall cycles of the dispatch
    d['mfma_util_of_1024_simds'] = d['SQ_VALU_MFMA_BUSY_CYCLES']['avg'] / (1024.0 * d['GRBM_GUI_ACTIVE']['avg'])
  # VALU / LDS side.  On this pool rocprofv3 reports the SQ counters PER SHADER ENGINE (SQ_WAVES of a
  # dispatch = its waves / 32): one engine = 8 CUs = 32 SIMDs.  SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES
  # count quad-cycles, SQ_BUSY_CYCLES the engine's busy cycles of the dispatch.
  busy = d.get('SQ_BUSY_CYCLES', {}).get('avg', 0)
  if 'SQ_ACTIVE_INST_VALU' in d and busy > 0:
    d['valu_busy_frac'] = 4.0 * d['SQ_ACTIVE_INST_VALU']['avg'] / (32.0 * busy)   # of the SIMDs' cycles
  if 'SQ_INSTS_VALU' in d and 'SQ_WAVES' in d and d['SQ_WAVES']['avg'] > 0:
    d['valu_insts_per_wave'] = d['SQ_INSTS_VALU']['avg'] / d['SQ_WAVES']['avg']
  if 'SQ_INSTS_LDS' in d and 'SQ_WAVES' in d and d['SQ_WAVES']['avg'] > 0:
    d['lds_insts_per_wave'] = d['SQ_INSTS_LDS']['avg'] / d['SQ_WAVES']['avg']
  if 'SQ_LDS_IDX_ACTIVE' in d and busy > 0:
    d['lds_active_frac'] = d['SQ_LDS_IDX_ACTIVE']['avg'] / (8.0 * busy)   # of the 8 CUs' LDS cycles
  if 'SQ_LDS_BANK_CONFLICT' in d and 'SQ_LDS_IDX_ACTIVE' in d and d['SQ_LDS_IDX_ACTIVE']['avg'] > 0:
    d['lds_conflict_over_active'] = d['SQ_LDS_BANK_CONFLICT']['avg'] / d['SQ_LDS_IDX_ACTIVE']['avg']
  if 'SQ_WAVE_CYCLES' in d and 'SQ_ACTIVE_INST_VALU' in d and d['SQ_WAVE_CYCLES']['avg'] > 0:
    # share of a wave's resident cycles in which one of its VALU instructions executes
    d['valu_active_over_wave_cycles'] = d['SQ_ACTIVE_INST_VALU']['avg'] / d['SQ_WAVE_CYCLES']['avg']
  if 'SQ_WAVE_CYCLES' in d and busy > 0:
    d['waves_per_simd_resident'] = 4.0 * d['SQ_WAVE_CYCLES']['avg'] / (32.0 * busy)
out = dict(res)
out['kernels'] = {k: {kk: vv for kk, vv in d.items() if not isinstance(vv, dict)} for k, d in res.items()}
if source:
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out['_meta'] = {'source_file': source,
                  'source_sha1': hashlib.sha1(open(os.path.join(root, source), 'rb').read()).hexdigest()}
json.dump(out, open(out_path, 'w'), indent=1, sort_keys=True)
for k, d in sorted(res.items()):
  print(k[:70].ljust(70), {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in d.items() if not isinstance(vv, dict)},
        {kk: '%.4g' % vv['avg'] for kk, vv in d.items() if isinstance(vv, dict) and kk.startswith(('SQ_', 'GRBM'))})
