"""The PMC summaries bench.py quotes (`roofline.traffic`, `valu_frac`) describe the kernels it runs:
each tracked summary carries the SHA-1 of the kernel source it was collected on, and bench.py drops the
figure when the source has changed since.  This CPU test fails as soon as a kernel file is edited
without re-collecting its counters (tools/gpu.sh pmc_conv / pmc_warp / pmc_warp_valu), so a stale
summary cannot reach the driver's bench line unnoticed."""
import hashlib
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SUMMARIES = [
    'profiles/r06_conv_pmc_1024_1024_3_1_32_64_1_8.json',
    'profiles/r06_warp_pmc.json',
    'profiles/r06_warp_valu_pmc.json',
]


@pytest.mark.parametrize('path', SUMMARIES)
def test_pmc_summary_matches_the_kernel_source(path):
  d = json.load(open(os.path.join(ROOT, path)))
  meta = d.get('_meta')
  assert meta and meta.get('source_file') and meta.get('source_sha1'), f'{path} carries no source hash'
  src = os.path.join(ROOT, meta['source_file'])
  sha = hashlib.sha1(open(src, 'rb').read()).hexdigest()
  assert sha == meta['source_sha1'], (
      f'{meta["source_file"]} changed since {path} was collected: re-run the PMC pass '
      '(tools/gpu.sh pmc_conv / pmc_warp / pmc_warp_valu) and copy the summary into profiles/')


def test_bench_reads_the_tracked_summaries():
  text = open(os.path.join(ROOT, 'bench.py')).read() + open(os.path.join(ROOT, 'se3ds_amd', 'bench_step.py')).read()
  for path in SUMMARIES:
    assert os.path.basename(path).split('.')[0][:12] in text, path
