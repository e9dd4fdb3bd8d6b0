"""Helpers shared by bench.py and se3ds_amd/bench_step.py: HBM traffic per launch from the tracked
rocprofv3 PMC summaries under profiles/ (PMC passes cannot run inside a bench process), with the
provenance the bench line needs: the numbers are NOT measured in this run, and they are dropped
(traffic = null) when the kernel source has changed since the PMC pass."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc_traffic(json_name, kernels):
  """HBM bytes per launch from the TRACKED rocprofv3 PMC summaries under profiles/ (separate
  FETCH_SIZE / WRITE_SIZE passes, tools/gpu_r3_prof.sh; raw counter values, the guide's x2 gfx950
  correction for wide coalesced reads is listed beside them).  None when the file is missing."""
  path = os.path.join(ROOT, 'profiles', json_name)
  if not os.path.exists(path):
    return None, None
  d = json.load(open(path))
  fetch = write = 0.0
  used = []
  for k, v in d.items():
    if any(t in k for t in kernels) and 'fetch_mb' in v and 'write_mb' in v:
      fetch += v['fetch_mb']
      write += v['write_mb']
      used.append(k)
  if not used:
    return None, None
  # MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports half the bytes of wide
  # coalesced reads (16 B / lane) -> doubled; WRITE_SIZE is taken as reported
  detail = {'source': 'profiles/' + json_name, 'measured_in_this_run': False,
            'source_sha1': d.get('_meta', {}).get('source_sha1'),
            'current_sha1': sha1(d.get('_meta', {}).get('source_file')),
            'kernels': sorted(used), 'fetch_mb_raw': fetch,
            'fetch_mb_x2_wide_read_correction': 2 * fetch, 'write_mb': write,
            'correction': 'traffic = 2 x FETCH_SIZE + WRITE_SIZE (guide: gfx950 wide-read x2)'}
  if detail['source_sha1'] is None or detail['source_sha1'] != detail['current_sha1']:
    detail['stale'] = 'kernel source changed since the PMC pass (or the summary carries no hash)'
    return None, detail
  return (2 * fetch + write) * 1e6, detail


def sha1(rel):
  import hashlib
  if not rel or not os.path.exists(os.path.join(ROOT, rel)):
    return None
  return hashlib.sha1(open(os.path.join(ROOT, rel), 'rb').read()).hexdigest()
