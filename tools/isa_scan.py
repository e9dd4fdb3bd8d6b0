"""Condensed instruction stream of one kernel from a -save-temps .s file:
python tools/isa_scan.py file.s kernel_substring [start_label]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + pat + r'\S*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
out = []
run = None; cnt = 0
def flush():
  global run, cnt
  if run: out.append(f'    {run} x{cnt}')
  run = None; cnt = 0
for i in range(start, end):
  l = lines[i].strip()
  if not l or l.startswith(';'): continue
  op = l.split()[0]
  key = None
  if re.match(r'\.LBB', l): flush(); out.append(l.split(';')[0]); continue
  if op.startswith('v_mfma'): key = 'mfma'
  elif op.startswith('ds_read') or op.startswith('ds_load'): key = op
  elif op.startswith('ds_write') or op.startswith('ds_store'): key = op
  elif 'lds' in l and (op.startswith('global_load') or op.startswith('buffer_load')): key = 'GLDS'
  elif op.startswith('global_load') or op.startswith('buffer_load'): key = op
  elif op.startswith('global_store') or op.startswith('buffer_store'): key = op
  elif op in ('s_waitcnt', 's_barrier', 's_setprio') or op.startswith('s_cbranch') or op == 's_branch' or op.startswith('scratch_'):
    flush(); out.append('  ' + l.split(';')[0].strip()); continue
  else: key = 'other'
  if key == run: cnt += 1
  else:
    flush(); run = key; cnt = 1
flush()
print('\n'.join(out))
