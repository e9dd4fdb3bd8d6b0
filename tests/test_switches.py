"""Every environment switch of the library is documented and owned.

A switch is any SE3DS_* name the sources READ from the environment (getenv / os.environ).  Each one
must (1) have a row in DESIGN.md section 9, and (2) be exercised by the test or tool this registry
names -- the owner file has to mention the switch (or the module attribute the switch sets, for the
in-process tests that flip the attribute instead of re-importing under another environment).
A switch without an owner is a path nobody runs: delete it or give it one."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# switch -> (owner file, token that must appear in the owner)
OWNERS = {
    # convolutions (csrc/conv.hip)
    'SE3DS_BIG_TILE': ('tests/test_nets_gpu.py', 'SE3DS_BIG_TILE'),
    'SE3DS_HALO_TILE': ('tests/test_nets_gpu.py', 'SE3DS_HALO_TILE'),
    'SE3DS_HALO_M16': ('tests/test_nets_gpu.py', 'SE3DS_HALO_M16'),
    'SE3DS_NO_THIN': ('tests/test_nets_gpu.py', 'SE3DS_NO_THIN'),
    'SE3DS_FUSED_BN_STATS': ('tests/test_nets_gpu.py', 'SE3DS_FUSED_BN_STATS'),
    'SE3DS_PROBE_MODE': ('tools/conv_phases.py', 'SE3DS_PROBE_MODE'),
    # normalisation / tape (csrc/norm.hip, hipops/nn.py)
    'SE3DS_NORM_CG': ('tests/test_blocks_gpu.py', 'SE3DS_NORM_CG'),
    'SE3DS_CONVT_2X2': ('tests/test_blocks_gpu.py', '_CONVT_2X2'),
    'SE3DS_MASK_CACHE': ('tests/test_blocks_gpu.py', '_MASK_CACHE'),
    'SE3DS_DROP_1X1_MASK': ('tests/test_blocks_gpu.py', '_DROP_1X1_MASK'),
    'SE3DS_FUSED_BN_BWD': ('tests/test_blocks_gpu.py', '_FUSED_BN_BWD'),
    'SE3DS_FUSED_ROW_SCALE': ('tests/test_blocks_gpu.py', '_FUSED_ROW_SCALE'),
    'SE3DS_NORM_DEBUG': ('tests/test_blocks_gpu.py', '_NORM_DEBUG'),
    'SE3DS_CHECK_MASKS': ('tests/test_nets_gpu.py', '_CHECK_MASKS'),
    # warp (csrc/geom.hip, utils/point_cloud_utils.py)
    'SE3DS_SPLAT_SCATTER': ('tests/test_warp_gpu.py', 'SE3DS_SPLAT_SCATTER'),
    'SE3DS_SPLAT_PACKED': ('tests/test_warp_gpu.py', 'SE3DS_SPLAT_PACKED'),
    'SE3DS_SPLAT_SORT': ('tests/test_warp_gpu.py', 'SE3DS_SPLAT_SORT'),
    'SE3DS_SPLAT_SLICE': ('tests/test_warp_gpu.py', 'SE3DS_SPLAT_SLICE'),
    'SE3DS_SPLAT_PTS': ('tests/test_warp_gpu.py', 'SE3DS_SPLAT_PTS'),
    'SE3DS_SPLAT_SUPERPX': ('tests/test_warp_gpu.py', 'SE3DS_SPLAT_SUPERPX'),
    'SE3DS_SPLAT_DEBUG': ('tests/test_warp_gpu.py', 'SE3DS_SPLAT_DEBUG'),
    'SE3DS_UNPROJECT_VEC': ('tests/test_warp_gpu.py', 'SE3DS_UNPROJECT_VEC'),
    'SE3DS_CHECK_PROMISE': ('tests/test_warp_gpu.py', '_CHECK_PROMISE_SYNC'),
    # the step's schedule (models/image_models.py, trainers/)
    'SE3DS_DUAL_STREAM': ('tools/step_compare.py', 'SE3DS_DUAL_STREAM'),
    'SE3DS_DUAL_PHASES': ('tools/step_compare.py', 'SE3DS_DUAL_PHASES'),
    'SE3DS_SEGMENT_OPTIMIZER': ('tools/step_compare.py', 'SE3DS_SEGMENT_OPTIMIZER'),
    'SE3DS_FUSED_CLIP_ADAM': ('tools/step_compare.py', 'SE3DS_FUSED_CLIP_ADAM'),
    'SE3DS_DEFER_WGRAD_REDUCE': ('tools/step_compare.py', 'SE3DS_DEFER_WGRAD_REDUCE'),
    'SE3DS_WGRAD_STREAM': ('tools/step_compare.py', 'SE3DS_WGRAD_STREAM'),
    'SE3DS_D_OVERLAP': ('tools/step_compare.py', 'SE3DS_D_OVERLAP'),
    'SE3DS_CU_MASK': ('tools/probes/cu_mask_ab.sh', 'SE3DS_CU_MASK'),
    'SE3DS_UNFUSED_EMA': ('tests/test_nets_gpu.py', 'SE3DS_UNFUSED_EMA'),
    # several replicas
    'SE3DS_DUAL_STREAM_DP': ('tests/test_dist_gpu.py', 'SE3DS_DUAL_STREAM_DP'),
    'SE3DS_FORCE_GRAD_SYNC': ('tests/test_nets_gpu.py', 'SE3DS_FORCE_GRAD_SYNC'),
    'SE3DS_GRAD_SYNC_OWN_COMM': ('tests/test_dist_gpu.py', 'SE3DS_GRAD_SYNC_OWN_COMM'),
    # bench
    'SE3DS_BENCH_BACKEND': ('tests/test_dist_gpu.py', 'SE3DS_BENCH_BACKEND'),
    'SE3DS_BENCH_SHAPES': ('tools/probes/shapes_ab.sh', 'SE3DS_BENCH_SHAPES'),
    # which build of the library a process loads (A/B of two builds on one box)
    'SE3DS_LIB': ('tools/probes/lib_ab_step.sh', 'SE3DS_LIB'),
}

_READ = re.compile(r'''(?:getenv\(\s*|sort_env\(\s*|environ\.get\(\s*|environ\[\s*)["'](SE3DS_[A-Z0-9_]+)["']''')


def _product_sources():
  for base, _, files in os.walk(os.path.join(ROOT, 'se3ds_amd')):
    if '_obj' in base or '__pycache__' in base:
      continue
    for f in files:
      if f.endswith(('.py', '.hip', '.h')):
        yield os.path.join(base, f)
  yield os.path.join(ROOT, 'bench.py')
  yield os.path.join(ROOT, '__graft_entry__.py')


def _switches_read():
  found = {}
  for path in _product_sources():
    for name in _READ.findall(open(path, errors='replace').read()):
      found.setdefault(name, []).append(os.path.relpath(path, ROOT))
  return found


def test_every_switch_the_sources_read_has_an_owner_and_every_owner_a_switch():
  found = _switches_read()
  assert found, 'the scan found no switch at all: the pattern is broken'
  missing = sorted(set(found) - set(OWNERS))
  assert not missing, f'switches without an owner: { {k: found[k] for k in missing} }'
  stale = sorted(set(OWNERS) - set(found))
  assert not stale, f'registry rows for switches no source reads any more: {stale}'


def test_every_owner_exercises_its_switch():
  for name, (owner, token) in OWNERS.items():
    path = os.path.join(ROOT, owner)
    assert os.path.exists(path), (name, owner)
    assert token in open(path).read(), f'{owner} does not mention {token} (owner of {name})'


def test_design_section_9_lists_every_switch():
  text = open(os.path.join(ROOT, 'DESIGN.md')).read()
  start = text.index('## 9. Environment switches')
  sec = text[start:]
  rows = set(re.findall(r'^\| `(SE3DS_[A-Z0-9_]+)`', sec, flags=re.M))
  assert rows == set(OWNERS), (sorted(set(OWNERS) - rows), sorted(rows - set(OWNERS)))
  for name, (owner, _) in OWNERS.items():
    row = next(l for l in sec.splitlines() if l.startswith(f'| `{name}`'))
    assert owner in row, f'DESIGN section 9 names another owner than the registry for {name}: {row}'
