"""GPU parity tests of the point-cloud warp: HIP path (through the C ABI) vs the CPU oracle,
bit-exact (indices, masks, depth, features)."""
import os

import numpy as np
import pytest
import torch

from oracle import warp_c
from oracle import warp_np
from se3ds_amd.utils import pano_utils
from se3ds_amd.utils import point_cloud_utils

pytestmark = pytest.mark.gpu
DEPTH_SCALE = 20.0
F32 = np.float32


def dev():
  assert torch.cuda.is_available(), 'GPU tests need an MI355X'
  return torch.device('cuda:0')


def t(a):
  return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def np_project(feats, xyz1, h, w, void, offset=None):
  """The INDEPENDENT statement of the render (oracle/warp_np.py: NumPy scatter-min / max, libm
  transcendentals -- it shares no arithmetic with the kernels, unlike the C twin warp_c, which includes
  the kernels' own se3ds_geom_math.h): (depth, feats) of project_feats_to_equirectangular on
  xyz1 - offset."""
  xyz1 = np.asarray(xyz1, F32)
  if offset is not None:
    n = xyz1.shape[0]
    xyz1 = (xyz1 - np.concatenate([np.asarray(offset, F32), np.zeros((n, 1), F32)], 1)[:, :, None]).astype(F32)
  return warp_np.project_feats_to_equirectangular(feats, xyz1, h, w, void, DEPTH_SCALE)


def synth_pano(rng, n, h, w):
  rgb = rng.integers(0, 256, (n, h, w, 3)).astype(np.int32)
  depth = rng.uniform(0, 1, (n, h, w)).astype(F32)
  poison = rng.uniform(0, 1, (n, h, w))
  depth[poison < 0.02] = 0.0
  depth[poison > 0.99] = 1.0
  return rgb, depth


@pytest.mark.parametrize('n,h,dtype', [(2, 16, np.int32), (1, 64, np.int32), (2, 32, np.uint8),
                                       (1, 32, np.float32)])
def test_unproject_bit_exact(n, h, dtype):
  rng = np.random.default_rng(10)
  w = 2 * h
  rgb, depth = synth_pano(rng, n, h, w)
  feats = rgb.astype(dtype)
  void = 0 if dtype == np.uint8 else -1
  pos = rng.standard_normal((n, 3)).astype(F32)
  xyz_o, f_o = warp_np.equirectangular_to_pointcloud(feats, depth, void, DEPTH_SCALE)
  xyz_g, f_g = pano_utils.equirectangular_to_pointcloud(t(feats), t(depth), void, DEPTH_SCALE)
  np.testing.assert_array_equal(xyz_g.cpu().numpy(), xyz_o)
  np.testing.assert_array_equal(f_g.cpu().numpy(), f_o)
  assert f_g.cpu().numpy().dtype == dtype
  # fused "+= position" (models.py:225-226)
  pos4 = np.concatenate([pos, np.zeros((n, 1), F32)], 1)[:, :, None]
  xyz_p, _ = pano_utils.equirectangular_to_pointcloud(t(feats), t(depth), void, DEPTH_SCALE,
                                                      position=t(pos))
  np.testing.assert_array_equal(xyz_p.cpu().numpy(), (xyz_o + pos4).astype(F32))
  # scalar feats path
  xs, fs = pano_utils.equirectangular_to_pointcloud(t(feats[..., 0]), t(depth), void, DEPTH_SCALE)
  assert fs.shape == (n, h * w)
  np.testing.assert_array_equal(fs.cpu().numpy(), f_o[..., 0])


@pytest.mark.parametrize('dtype,c', [(np.int32, 3), (np.float32, 3), (np.int32, 1)])
def test_unproject_vec4_kernel_equals_scalar_kernel(dtype, c, monkeypatch):
  """Round 5: four pixels per thread with 16-byte accesses (int32 / fp32 features, 1 or 3 channels,
  aligned windows) against the one-pixel-per-lane kernel (SE3DS_UNPROJECT_VEC=0) and the oracle,
  bit for bit -- also into an aligned window of a larger memory and with a position."""
  rng = np.random.default_rng(21)
  n, h = 2, 64
  w = 2 * h
  rgb, depth = synth_pano(rng, n, h, w)
  feats = rgb[..., :c].astype(dtype)
  pos = rng.standard_normal((n, 3)).astype(F32)
  xyz_o, f_o = warp_np.equirectangular_to_pointcloud(feats, depth, -1, DEPTH_SCALE)
  pos4 = np.concatenate([pos, np.zeros((n, 1), F32)], 1)[:, :, None]
  res = {}
  for vec in ('1', '0'):
    monkeypatch.setenv('SE3DS_UNPROJECT_VEC', vec)
    x, f = pano_utils.equirectangular_to_pointcloud(t(feats), t(depth), -1, DEPTH_SCALE, position=t(pos))
    np.testing.assert_array_equal(x.cpu().numpy(), (xyz_o + pos4).astype(F32))
    np.testing.assert_array_equal(f.cpu().numpy(), f_o)
    # a window of a larger memory: offset 4 * 37 (aligned) and 4 * 37 + 1 (scalar fallback)
    for off in (148, 149):
      m = 3 * h * w
      mem_x = torch.full((n, 4, m), 7.0, device=dev())
      mem_f = torch.full((n, m, c), 5, dtype=t(feats).dtype, device=dev())
      pano_utils.equirectangular_to_pointcloud(t(feats), t(depth), -1, DEPTH_SCALE, out=(mem_x, mem_f, off))
      np.testing.assert_array_equal(mem_x[:, :, off:off + h * w].cpu().numpy(), xyz_o)
      np.testing.assert_array_equal(mem_f[:, off:off + h * w].cpu().numpy(), f_o)
      assert float(mem_x[:, :, :off].min()) == 7.0 and float(mem_x[:, :, off + h * w:].min()) == 7.0
      assert int(mem_f[:, :off].min()) == 5 and int(mem_f[:, off + h * w:].min()) == 5
      # SE3DS_XYZ1_ONES_PRESET (round 6): the homogeneous row of the window is the CALLER's -- a point-cloud
      # memory fills it with 1.0 once -- and is not written; rows 0-2 and the features are as before
      mem_x2 = torch.full((n, 4, m), 7.0, device=dev())
      mem_f2 = torch.full((n, m, c), 5, dtype=t(feats).dtype, device=dev())
      pano_utils.equirectangular_to_pointcloud(t(feats), t(depth), -1, DEPTH_SCALE, out=(mem_x2, mem_f2, off),
                                               ones_preset=True)
      assert torch.equal(mem_x2[:, :3], mem_x[:, :3]) and torch.equal(mem_f2, mem_f)
      assert float(mem_x2[:, 3].min()) == 7.0 and float(mem_x2[:, 3].max()) == 7.0
    res[vec] = (x.cpu().numpy(), f.cpu().numpy())
  np.testing.assert_array_equal(res['0'][0], res['1'][0])
  np.testing.assert_array_equal(res['0'][1], res['1'][1])


@pytest.mark.parametrize('n,h,views', [(1, 64, 2), (2, 32, 3), (1, 256, 2)])
def test_trajectory_step_in_one_call_equals_the_separate_calls(n, h, views):
  """PointCloudMemory.append_views_and_project (se3ds_warp_views_to_target: V unprojects + one
  render queued by ONE library call) against append_equirect x V + project, and against the C
  oracle: bit-exact depth / features / mask, identical memory."""
  rng = np.random.default_rng(33)
  w = 2 * h
  vs = []
  for _ in range(views):
    rgb, depth = synth_pano(rng, n, h, w)
    pos = (rng.standard_normal((n, 3)) * 0.4).astype(F32)
    vs.append((rgb, depth, pos))
  tgt = (rng.standard_normal((n, 3)) * 0.4).astype(F32)
  g = [(t(r), t(d), t(p)) for r, d, p in vs]
  a = point_cloud_utils.PointCloudMemory(n, 3, torch.int32, dev())
  da, fa, ma = a.append_views_and_project(g, -1, DEPTH_SCALE, t(tgt), h, w, with_mask=True)
  b = point_cloud_utils.PointCloudMemory(n, 3, torch.int32, dev())
  for r, d, p in g:
    b.append_equirect(r, d, -1, DEPTH_SCALE, position=p)
  db, fb, mb = b.project(h, w, -1, DEPTH_SCALE, position=t(tgt), with_mask=True)
  assert a.m == b.m == views * h * w
  assert torch.equal(a.coords, b.coords) and torch.equal(a.feats, b.feats)
  assert torch.equal(da, db) and torch.equal(fa, fb) and torch.equal(ma, mb)
  # ... and the oracle (C twin of the kernels' arithmetic)
  tabs = warp_np.equirect_angle_tables(h, w)
  xs, fs = [], []
  for r, d, p in vs:
    x, f = warp_c.unproject_equirect(r, d, tabs, -1, DEPTH_SCALE, position=p)
    xs.append(x)
    fs.append(f)
  d_o, f_o = warp_c.project_feats_to_equirectangular(np.concatenate(fs, 1), np.concatenate(xs, 2), h, w,
                                                     -1, DEPTH_SCALE, offset=tgt)
  np.testing.assert_array_equal(da.cpu().numpy(), d_o)
  np.testing.assert_array_equal(fa.cpu().numpy(), f_o)
  # ... and the independent NumPy / libm statement of both halves
  xs_n, fs_n = [], []
  for r, d, p in vs:
    x, f = warp_np.equirectangular_to_pointcloud(r, d, -1, DEPTH_SCALE)
    xs_n.append((x + np.concatenate([p, np.zeros((n, 1), F32)], 1)[:, :, None]).astype(F32))
    fs_n.append(f)
  np.testing.assert_array_equal(a.coords.cpu().numpy(), np.concatenate(xs_n, 2))
  np.testing.assert_array_equal(a.feats.cpu().numpy(), np.concatenate(fs_n, 1))
  d_n, f_n = np_project(np.concatenate(fs_n, 1), np.concatenate(xs_n, 2), h, w, -1, offset=tgt)
  np.testing.assert_array_equal(da.cpu().numpy(), d_n)
  np.testing.assert_array_equal(fa.cpu().numpy(), f_n)
  # a second step appends behind the first and renders everything
  rgb, depth = synth_pano(rng, n, h, w)
  d2, f2 = a.append_views_and_project([(t(rgb), t(depth), None)], -1, DEPTH_SCALE, t(tgt), h, w)
  b.append_equirect(t(rgb), t(depth), -1, DEPTH_SCALE)
  d3, f3 = b.project(h, w, -1, DEPTH_SCALE, position=t(tgt))
  assert a.m == b.m and torch.equal(d2, d3) and torch.equal(f2, f3)
  # rendering into caller-owned frame buffers (out=): the same values, the same tensors back
  frame = (torch.empty_like(da), torch.empty_like(fa), torch.empty_like(ma))
  c = point_cloud_utils.PointCloudMemory(n, 3, torch.int32, dev())
  dc, fc, mc = c.append_views_and_project(g, -1, DEPTH_SCALE, t(tgt), h, w, with_mask=True, out=frame)
  assert dc is frame[0] and fc is frame[1] and mc is frame[2]
  assert torch.equal(dc, da) and torch.equal(fc, fa) and torch.equal(mc, ma)
  with pytest.raises(ValueError):
    c.append_views_and_project(g, -1, DEPTH_SCALE, t(tgt), h, w, out=(frame[0][:, :1], frame[1], None))


def test_unproject_errors():
  d = dev()
  with pytest.raises(ValueError):
    pano_utils.equirectangular_to_pointcloud(torch.zeros((2, 4), device=d),
                                             torch.zeros((1, 2, 4), device=d), 0, 20.0)
  with pytest.raises(ValueError):
    pano_utils.equirectangular_to_pointcloud(
        torch.zeros((1, 2, 4, 3), dtype=torch.uint8, device=d), torch.zeros((1, 2, 4), device=d),
        -1, 20.0)
  with pytest.raises(AssertionError):
    pano_utils.equirectangular_to_pointcloud(torch.zeros((1, 4, 4, 3), device=d),
                                             torch.zeros((1, 4, 4), device=d), 0, 20.0)


@pytest.mark.parametrize('n,h,views', [(1, 16, 1), (2, 32, 2), (1, 128, 2), (1, 256, 1)])
def test_project_equirect_bit_exact(n, h, views, monkeypatch):
  monkeypatch.setenv('SE3DS_SPLAT_DEBUG', '1')   # (the packed / sorted kernels write the (idx, z) parity tap on request)
  rng = np.random.default_rng(11 + h)
  w = 2 * h
  coords, feats = [], []
  for _ in range(views):
    rgb, depth = synth_pano(rng, n, h, w)
    xyz1, f = warp_np.equirectangular_to_pointcloud(rgb, depth, -1, DEPTH_SCALE)
    pos = (rng.standard_normal((n, 3)) * 0.5).astype(F32)
    xyz1 = xyz1 + np.concatenate([pos, np.zeros((n, 1), F32)], 1)[:, :, None]
    coords.append(xyz1.astype(F32))
    feats.append(f)
  mem_xyz = np.concatenate(coords, 2)
  mem_rgb = np.concatenate(feats, 1)
  target = (rng.standard_normal((n, 3)) * 0.5).astype(F32)
  m = mem_xyz.shape[2]

  d_o, f_o = warp_c.project_feats_to_equirectangular(mem_rgb, mem_xyz, h, w, -1, DEPTH_SCALE,
                                                     offset=target)
  d_g, f_g, m_g = pano_utils.project_feats_to_equirectangular(
      t(mem_rgb), t(mem_xyz), h, w, -1, DEPTH_SCALE, offset=t(target), with_mask=True)
  # index-level parity: first-stage flat indices of every point
  rel = (mem_xyz - np.concatenate([target, np.zeros((n, 1), F32)], 1)[:, :, None]).astype(F32)
  proj = warp_c.equirect_project_coords(rel)
  _, _, flat = warp_c.project_to_feat(proj, mem_rgb, h, w, DEPTH_SCALE, -1, return_flat=True)
  idx_g, z_g = point_cloud_utils.splat_debug_indices(n, m, dev())
  idx_g = idx_g.cpu().numpy().astype(np.int64)
  flat_g = np.where(idx_g < 0, 0, idx_g + (np.arange(n)[:, None] * h * w)).reshape(-1)
  np.testing.assert_array_equal(flat_g, flat)
  np.testing.assert_array_equal(z_g.cpu().numpy(), proj[:, 2])
  np.testing.assert_array_equal(d_g.cpu().numpy(), d_o)
  np.testing.assert_array_equal(f_g.cpu().numpy(), f_o)
  np.testing.assert_array_equal(m_g.cpu().numpy()[..., None], warp_np.proj_mask(d_o, f_o, -1))
  # the NumPy statement (libm transcendentals) agrees too
  d_n, f_n = warp_np.project_feats_to_equirectangular(mem_rgb, rel, h, w, -1, DEPTH_SCALE)
  np.testing.assert_array_equal(d_g.cpu().numpy(), d_n)
  np.testing.assert_array_equal(f_g.cpu().numpy(), f_n)


def test_roundtrip_property_full_size():
  """512x1024 (BASELINE config size): unproject then project at the same position returns
  the panorama on >= 95 % of pixels (models_test.py:62-68); we expect every valid pixel."""
  rng = np.random.default_rng(12)
  h, w = 512, 1024
  rgb, depth = synth_pano(rng, 1, h, w)
  pos = t(rng.standard_normal((1, 3)).astype(F32))
  xyz1, f = pano_utils.equirectangular_to_pointcloud(t(rgb), t(depth), -1, DEPTH_SCALE,
                                                     position=pos)
  pd, prgb, pm = pano_utils.project_feats_to_equirectangular(f, xyz1, h, w, -1, DEPTH_SCALE,
                                                             offset=pos, with_mask=True)
  valid = (depth > 0) & (depth < 1)
  eq = torch.all(prgb.to(torch.int32) == t(rgb), dim=-1).cpu().numpy()
  assert eq[valid].mean() > 0.999
  assert eq.mean() >= 0.95
  got_d = pd.cpu().numpy()
  # depth comes back as |xyz| / depth_scale: equal to the input up to fp32 rounding
  ok = np.isclose(got_d[valid], depth[valid], rtol=1e-4, atol=2e-6)
  assert ok.mean() > 0.999
  # idempotence: re-unprojecting the projection and projecting again is a fixed point
  xyz2, f2 = pano_utils.equirectangular_to_pointcloud(prgb.to(torch.int32), pd, -1, DEPTH_SCALE)
  pd2, prgb2 = pano_utils.project_feats_to_equirectangular(f2, xyz2, h, w, -1, DEPTH_SCALE)
  m = pm.cpu().numpy() > 0
  m[0, 0, 0] = False  # the sink pixel
  assert np.isclose(pd2.cpu().numpy()[m], got_d[m], rtol=1e-5, atol=1e-7).mean() > 0.999
  assert torch.all(prgb2[torch.from_numpy(m).to(prgb2.device)] == prgb[torch.from_numpy(m).to(prgb.device)])


def test_project_wide_features_take_the_scatter_path():
  """More than 7 channels do not fit the binned splat's LDS tile: the global-atomic scatter
  version must stay bit-exact too."""
  rng = np.random.default_rng(29)
  n, h, w, m, c = 2, 32, 64, 6000, 9
  xyz = rng.standard_normal((n, 4, m)).astype(F32) * 3.0
  feats = rng.integers(0, 200, (n, m, c)).astype(np.int32)
  feats[rng.uniform(size=(n, m)) < 0.05] = -1   # void points go to the sink
  d_o, f_o = warp_c.project_feats_to_equirectangular(feats, xyz, h, w, -1, DEPTH_SCALE)
  d_g, f_g = pano_utils.project_feats_to_equirectangular(t(feats), t(xyz), h, w, -1, DEPTH_SCALE)
  np.testing.assert_array_equal(d_g.cpu().numpy(), d_o)
  np.testing.assert_array_equal(f_g.cpu().numpy(), f_o)
  d_n, f_n = np_project(feats, xyz, h, w, -1)
  np.testing.assert_array_equal(d_g.cpu().numpy(), d_n)
  np.testing.assert_array_equal(f_g.cpu().numpy(), f_n)


def test_project_edge_cases():
  d = dev()
  h, w = 8, 16
  # M = 0 (VLN notebook passes an all-void semantic memory)
  pd, pf = pano_utils.project_feats_to_equirectangular(
      torch.zeros((1, 0, 1), dtype=torch.uint8, device=d), torch.zeros((1, 4, 0), device=d), h, w,
      0, DEPTH_SCALE)
  assert torch.all(pd == 1.0) and torch.all(pf == 0)
  # all points invalid (void feats) -> everything in the sink; depth[0,0,0] = min z
  rng = np.random.default_rng(13)
  xyz = rng.standard_normal((2, 4, 50)).astype(F32)
  feats = np.full((2, 50, 3), -1, np.int32)
  d_o, f_o = warp_c.project_feats_to_equirectangular(feats, xyz, h, w, -1, DEPTH_SCALE)
  d_g, f_g = pano_utils.project_feats_to_equirectangular(t(feats), t(xyz), h, w, -1, DEPTH_SCALE)
  np.testing.assert_array_equal(d_g.cpu().numpy(), d_o)
  np.testing.assert_array_equal(f_g.cpu().numpy(), f_o)
  d_n, f_n = np_project(feats, xyz, h, w, -1)   # (the independent statement, here and below)
  np.testing.assert_array_equal(d_g.cpu().numpy(), d_n)
  np.testing.assert_array_equal(f_g.cpu().numpy(), f_n)
  # heavy collisions: all points in a few pixels, scalar uint8 semantic feats
  xyz = (rng.standard_normal((1, 4, 5000)) * 0.01 + np.array([1, 1, 1, 0])[None, :, None]).astype(F32)
  sem = rng.integers(0, 42, (1, 5000)).astype(np.uint8)
  d_o, f_o = warp_c.project_feats_to_equirectangular(sem, xyz, h, w, 0, DEPTH_SCALE)
  d_g, f_g = pano_utils.project_feats_to_equirectangular(t(sem), t(xyz), h, w, 0, DEPTH_SCALE)
  np.testing.assert_array_equal(d_g.cpu().numpy(), d_o)
  np.testing.assert_array_equal(f_g.cpu().numpy(), f_o)
  d_n, f_n = np_project(sem, xyz, h, w, 0)
  np.testing.assert_array_equal(d_g.cpu().numpy(), d_n)
  np.testing.assert_array_equal(f_g.cpu().numpy(), f_n)
  with pytest.raises(ValueError):
    pano_utils.project_feats_to_equirectangular(torch.zeros((1, 2, 3, 4), device=d),
                                                torch.zeros((1, 4, 2), device=d), h, w, 0, 20.0)


@pytest.mark.parametrize('n,size,multi', [(2, 64, False), (1, 128, True)])
def test_perspective_unproject_and_project_to_feat(n, size, multi):
  # utils/point_cloud_utils_test.py:26-64, against the oracle
  rng = np.random.default_rng(14)
  shape = (n, size, size) + ((3,) if multi else ())
  feats = rng.integers(0, 42, shape).astype(np.int32)
  depth = rng.uniform(0, 1.2, (n, size, size)).astype(F32)
  x_o, f_o = warp_np.get_filtered_coords_and_feats(feats, depth, DEPTH_SCALE)
  x_g, f_g = point_cloud_utils.get_filtered_coords_and_feats(t(feats), t(depth), DEPTH_SCALE)
  np.testing.assert_allclose(x_g.cpu().numpy(), x_o, rtol=2e-7, atol=1e-7)  # 4-term fp32 matmul
  np.testing.assert_array_equal(f_g.cpu().numpy(), f_o)
  xg = x_g.cpu().numpy()
  d_o, p_o = warp_c.project_to_feat(xg, f_o, size, size, DEPTH_SCALE, 0)
  d_g, p_g = point_cloud_utils.project_to_feat(x_g, f_g, size, size, DEPTH_SCALE, 0)
  np.testing.assert_array_equal(d_g.cpu().numpy(), d_o)
  np.testing.assert_array_equal(p_g.cpu().numpy(), p_o)
  assert float(d_g.min()) >= 0 and float(d_g.max()) <= 1


def test_project_to_feat_negative_output_void_and_negative_z():
  rng = np.random.default_rng(15)
  h, w = 8, 8
  xyz = rng.standard_normal((2, 4, 400)).astype(F32)  # many z < 0 -> sink gets negative z
  feats = (rng.standard_normal((2, 400, 2)) * 3).astype(F32)
  d_o, f_o = warp_c.project_to_feat(xyz, feats, h, w, DEPTH_SCALE, -100.0, -5.0)
  d_g, f_g = point_cloud_utils.project_to_feat(t(xyz), t(feats), h, w, DEPTH_SCALE, -100.0, -5.0)
  np.testing.assert_array_equal(d_g.cpu().numpy(), d_o)
  np.testing.assert_array_equal(f_g.cpu().numpy(), f_o)


def test_plane_known_answer_on_gpu():
  # models/models_test.py:81-137 through the product path
  from tests.test_oracle_warp import _plane_depth
  rng = np.random.default_rng(0)
  size = 4
  rgb = rng.integers(0, 255, (2, size, size * 2, 3)).astype(np.int32)
  depth = _plane_depth(size)
  pos = np.array([[0, 0, 0], [1, 0, 0]], F32)
  xyz1, feats = pano_utils.equirectangular_to_pointcloud(t(rgb), t(depth), -1, DEPTH_SCALE,
                                                         interpolation_method='bilinear',
                                                         position=t(pos))
  pc, mem = pano_utils.compact_valid_points(xyz1, feats, -1)
  assert tuple(pc.shape) == (2, 4, 24)
  pc, mem = pc.cpu().numpy(), mem.cpu().numpy()
  for ix, (axis, value) in enumerate([(1, 1), (0, 2)]):
    valid = np.any(mem[ix] != -1, axis=1)
    np.testing.assert_allclose(pc[ix][axis][valid], size**2 * [value], rtol=1e-6, atol=1e-6)


def test_compaction_matches_oracle():
  rng = np.random.default_rng(16)
  n, m = 2, 10007
  xyz = rng.standard_normal((n, 4, m)).astype(F32)
  feats = rng.integers(-1, 3, (n, m, 3)).astype(np.int32)
  feats[:, rng.uniform(size=m) < 0.4] = -1
  x_o, f_o = warp_np.compact_valid(xyz, feats, -1)
  x_g, f_g = pano_utils.compact_valid_points(t(xyz), t(feats), -1)
  np.testing.assert_array_equal(x_g.cpu().numpy(), x_o)
  np.testing.assert_array_equal(f_g.cpu().numpy(), f_o)


def test_mask_pano_gpu():
  rng = np.random.default_rng(17)
  for dtype in (np.float32, np.int32, np.uint8):
    pano = rng.integers(1, 255, (2, 64, 128, 3)).astype(dtype)
    for val in (0, -1) if dtype != np.uint8 else (0,):
      np.testing.assert_array_equal(pano_utils.mask_pano(t(pano), masked_region_value=val).cpu().numpy(),
                                    warp_np.mask_pano(pano, masked_region_value=val))
  assert pano_utils.crop_pano(t(pano)).shape == (2, 48, 128, 3)


def test_pixel_rays_golden_product(golden_dir):
  g = np.load(os.path.join(golden_dir, 'reference_literals.npz'))['pixel_rays_3']
  rays = pano_utils.equirectangular_pixel_rays(3).numpy().T.reshape(3, 6, 3)
  np.testing.assert_allclose(rays, g, rtol=1e-6, atol=1e-6)


def test_bilinear_and_resampling_paths():
  rng = np.random.default_rng(18)
  grid = rng.standard_normal((2, 17, 23, 3)).astype(F32)
  q = (rng.uniform(-2, 25, (2, 500, 2))).astype(F32)
  for indexing in ('ij', 'xy'):
    o = warp_np.interpolate_bilinear(grid, q, indexing)
    g = pano_utils.interpolate_bilinear(t(grid), t(q), indexing).cpu().numpy()
    np.testing.assert_array_equal(g, o)
  # rotate_pano (parity unpinned by the reference; oracle follows tfa's published algorithm)
  h, w = 32, 64
  pano = rng.uniform(0, 1, (2, h, w, 3)).astype(F32)
  def rot(a, b):
    ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
    return (np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]]) @
            np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])).astype(F32)
  mats = np.stack([rot(0.3, -1.1), rot(-0.2, 2.0)])
  c_o = warp_np.rotate_coords(mats, h, w, h)
  r_o = warp_np.rotate_pano(pano, mats)
  r_g = pano_utils.rotate_pano(t(pano), t(mats)).cpu().numpy()
  # 3x3 fp32 products may round differently (sum order / contraction is not pinned by the
  # reference): coordinates agree to ~1e-5 px except on the heading seam (atan2 = +-pi), where a
  # 1-ulp difference moves the sample by a whole panorama width.  Random-noise panorama =>
  # require agreement on all but a handful of seam pixels.
  ok = np.isfinite(c_o).all(-1).reshape(2, h, w)
  close = np.abs(r_g - r_o).max(-1) <= 1e-4
  assert close[ok].mean() > 0.995, close[ok].mean()
  # perspective <-> equirect
  img = rng.uniform(0, 1, (24, 24, 3)).astype(F32)
  fov = np.array([np.pi / 2, np.pi / 2], F32)
  p_o = warp_np.project_perspective_image(img, fov, 16, rotations=np.array([0.1, 0.4], F32))
  p_g = pano_utils.project_perspective_image(t(img), fov, 16, rotations=np.array([0.1, 0.4], F32))
  assert (np.abs(p_g.cpu().numpy() - p_o).max(-1) <= 1e-4).mean() > 0.995
  K = np.array([[12, 0, 11.5], [0, 12, 11.5], [0, 0, 1]], F32)
  e_o = warp_np.get_perspective_from_equirectangular_image(pano[0], K, mats[0], 24, 24)
  e_g = pano_utils.get_perspective_from_equirectangular_image(t(pano[0]), K, mats[0], 24, 24)
  assert (np.abs(e_g.cpu().numpy() - e_o).max(-1) <= 1e-4).mean() > 0.995


_OLD = dict(SE3DS_SPLAT_PACKED='0')   # the 20-byte-record paths behind the 8-byte packed default
_SORT = dict(SE3DS_SPLAT_SORT='2')    # round 4: sorted chunks + gathering resolve, also on small images
@pytest.mark.parametrize('env_extra', [
    dict(SE3DS_SPLAT_SLICE='48'),                          # packed records, every tile banded
    dict(_OLD, SE3DS_SPLAT_SLICE='48'),                    # three-pass, 20-byte records, every tile banded
    dict(_SORT, SE3DS_SPLAT_PTS='16'),                     # sorted chunks (16 points / thread) + gathering resolve
    dict(_SORT, SE3DS_SPLAT_PTS='8'),                      # ... 8 points / thread (4096-point chunks)
    dict(_SORT, SE3DS_SPLAT_PTS='4'),                      # ... 4 points / thread (2048-point chunks: what 512 x 1024 takes)
    dict(_SORT, SE3DS_SPLAT_SUPERPX='4096'),               # ... supertiles of 4096 pixels (two rows at 1024 x 2048)
    dict(_SORT, SE3DS_SPLAT_SUPERPX='512'),                # ... of 512 pixels (small images: column strips)
], ids=['packed-banded', 'three-pass-banded', 'sorted-16pt', 'sorted-8pt', 'sorted-4pt', 'sorted-4096px',
        'sorted-512px'])
def test_splat_banded_tiles_bit_exact(env_extra):
  """The splat parity tests re-run in a child process under switches that are read once per
  process: tiny slices (every tile of the small parity images is cut into bands of rows, in the
  packed and in the 20-byte-record resolve kernels), the 20-byte-record three-pass path that the
  8-byte packed path replaced as the default (SE3DS_SPLAT_PACKED=0; still what float features
  and more than 3 channels take); and the round-4 sorted-chunk path (SE3DS_SPLAT_SORT=2 takes it on
  images of any size) in its template variants.  (Round 2's opt-in single-pass binning kernel left
  the library in round 5.)"""
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, PYTHONPATH=root, **env_extra)
  env.pop('SE3DS_SPLAT_SCATTER', None)
  r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_warp_gpu.py'),
                      '-q', '-x', '-m', 'gpu',
                      # (round 6: also the uint8 / 1-2 channel / ragged-tail cases and the trajectory step,
                      #  so that every template variant of the sort kernel meets them)
                      '-k', ('(project or trajectory) and not banded' if 'SE3DS_SPLAT_PACKED' in env_extra else
                             '(project or packed_splat or trajectory) and not banded')],   # (the promise verdict is the packed paths')
                     env=env, cwd=root, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
  assert ' passed' in r.stdout


def test_packed_splat_variants_and_byte_range_promise():
  """The 8-byte packed-record splat (se3ds_amd/csrc/geom.hip, round 3) on the inputs that decide
  its dispatch: point counts that are not a multiple of 4 (scalar loads, padded tails), 1 / 2 / 3
  channels, uint8 (packed by type) and int32 (packed under the byte-range promise, which the
  wrapper establishes with se3ds_feats_byte_range and caches on the tensor); int32 features
  OUTSIDE [0, 255] must take the 20-byte path and stay exact; a broken promise at the C ABI is
  reported by se3ds_splat_promise_broken."""
  from se3ds_amd import _lib
  rng = np.random.default_rng(31)
  h, w = 48, 96
  for m, c, dt, void, hi_val in [(70001, 3, np.int32, -1, 256), (4099, 1, np.uint8, 0, 42),
                                 (12345, 2, np.int32, -1, 256), (8192, 3, np.uint8, 0, 256),
                                 (30000, 3, np.int32, -1, 100000)]:
    xyz = (rng.standard_normal((1, 4, m)) * rng.uniform(0.05, 6, (1, 1, m))).astype(F32)
    feats = rng.integers(0, hi_val, (1, m, c)).astype(dt)
    feats[rng.uniform(size=(1, m)) < 0.05] = void
    off = (rng.standard_normal((1, 3)) * 0.3).astype(F32)
    tf = t(feats)
    want_packable = dt == np.uint8 or hi_val <= 256
    assert point_cloud_utils.byte_range(tf, void) == want_packable
    d_o, f_o = warp_c.project_feats_to_equirectangular(feats, xyz, h, w, void, DEPTH_SCALE, offset=off)
    d_g, f_g, m_g = pano_utils.project_feats_to_equirectangular(tf, t(xyz), h, w, void, DEPTH_SCALE,
                                                                 offset=t(off), with_mask=True,
                                                                 mask_void=0)
    np.testing.assert_array_equal(d_g.cpu().numpy(), d_o, err_msg=str((m, c, dt)))
    np.testing.assert_array_equal(f_g.cpu().numpy(), f_o, err_msg=str((m, c, dt)))
    d_n, f_n = np_project(feats, xyz, h, w, void, offset=off)   # (independent NumPy / libm statement)
    np.testing.assert_array_equal(d_g.cpu().numpy(), d_n, err_msg=str((m, c, dt)))
    np.testing.assert_array_equal(f_g.cpu().numpy(), f_n, err_msg=str((m, c, dt)))
  # the cache follows torch's version counter: an in-place edit re-runs the check
  tf = t(rng.integers(0, 256, (1, 1000, 3)).astype(np.int32))
  assert point_cloud_utils.byte_range(tf, -1)
  tf[0, 5, 1] = 300
  assert not point_cloud_utils.byte_range(tf, -1)
  # a broken promise at the C ABI is visible afterwards (outputs of that call are undefined)
  m = 5000
  xyz = t((rng.standard_normal((1, 4, m)) * 3).astype(F32))
  L = _lib.lib()
  ws = torch.empty(L.se3ds_splat_workspace_bytes(1, m, h, w, 3), dtype=torch.uint8, device=dev())
  depth = torch.empty((1, h, w), device=dev())
  out = torch.empty((1, h, w, 3), device=dev())
  flag = torch.zeros(1, dtype=torch.int32, device=dev())
  for feats, want in ((rng.integers(0, 256, (1, m, 3)), 0), (rng.integers(0, 5000, (1, m, 3)), 1)):
    f = t(feats.astype(np.int32))
    rc = L.se3ds_project_equirect(_lib.ptr(xyz), None, _lib.ptr(f), _lib.I32 | point_cloud_utils.FEAT_BYTE_RANGE,
                                  1, m, 3, h, w, DEPTH_SCALE, -1.0, 0.0, _lib.ptr(depth), _lib.ptr(out),
                                  None, -1.0, _lib.ptr(ws), ws.numel(), _lib.stream())
    assert rc == 0
    assert L.se3ds_splat_promise_broken(_lib.ptr(ws), 1, m, _lib.ptr(flag), _lib.stream()) == 0
    assert int(flag.item()) == want
  # ... and STICKY in header word 3 of a workspace whose header the caller zeroed once: the clean
  # call that follows a violation does not hide it; reading with clear = 1 resets it
  ws[:256].zero_()
  seq = [rng.integers(0, 5000, (1, m, 3)), rng.integers(0, 256, (1, m, 3))]
  for feats in seq:
    f = t(feats.astype(np.int32))
    assert L.se3ds_project_equirect(_lib.ptr(xyz), None, _lib.ptr(f), _lib.I32 | point_cloud_utils.FEAT_BYTE_RANGE,
                                    1, m, 3, h, w, DEPTH_SCALE, -1.0, 0.0, _lib.ptr(depth), _lib.ptr(out),
                                    None, -1.0, _lib.ptr(ws), ws.numel(), _lib.stream()) == 0
  for want in (1, 0):
    assert L.se3ds_splat_promise_sticky(_lib.ptr(ws), _lib.ptr(flag), 1, _lib.stream()) == 0
    assert int(flag.item()) == want
  # the cached verdict belongs to ONE void class: all-(-1)-or-byte features are packable
  # for void -1 and NOT for void 0 (their -1 entries are then ordinary, unpackable values)
  tf = t(np.where(rng.uniform(size=(1, 2000, 3)) < 0.1, -1, rng.integers(0, 256, (1, 2000, 3))).astype(np.int32))
  assert point_cloud_utils.byte_range(tf, -1)
  assert not point_cloud_utils.byte_range(tf, 0)
  assert point_cloud_utils.byte_range(tf, -1)
  # bounds known by construction answer without reading the tensor back
  point_cloud_utils.set_int_range(tf, -1, 255)
  assert point_cloud_utils.byte_range(tf, -1)
  # the wrapper reports a promise broken behind its back (a raw-pointer writer that forgot
  # set_byte_range): the sticky flag is polled after promised splats
  xyz_np = (rng.standard_normal((1, 4, m)) * 3).astype(F32)
  good = t(rng.integers(0, 256, (1, m, 3)).astype(np.int32))
  assert point_cloud_utils.byte_range(good, -1)
  # (a raw-pointer write, no torch version bump: the float 3e9 reads back as the int32 1 328 730 206)
  import se3ds_amd.hipops  # noqa: F401  (registers se3ds_fill's signature)
  assert L.se3ds_fill(good.data_ptr() + 4 * 40, _lib.F32, 1, 3.0e9, _lib.stream()) == 0
  assert point_cloud_utils.byte_range(good, -1)   # the stale cached verdict
  # a FORCED poll (check_promise: the sync points of SE3DSModel / the roll-out, atexit) reports the
  # violation of a single promised splat -- a short trajectory never reaches the 64th
  try:
    point_cloud_utils.check_promise(dev())   # (drains whatever earlier tests left pending)
  except point_cloud_utils.PromiseBroken:
    pass
  pano_utils.project_feats_to_equirectangular(good, t(xyz_np), h, w, -1, DEPTH_SCALE)
  with pytest.raises(point_cloud_utils.PromiseBroken):
    point_cloud_utils.check_promise(dev())
  point_cloud_utils.check_promise(dev())   # (the flag was cleared by the read that raised)
  # ... and the non-blocking form raises one call late at most
  pano_utils.project_feats_to_equirectangular(good, t(xyz_np), h, w, -1, DEPTH_SCALE)
  with pytest.raises(point_cloud_utils.PromiseBroken):
    for _ in range(3):
      point_cloud_utils.check_promise(dev(), wait=False)
      torch.cuda.synchronize()
  old_every, point_cloud_utils._PROMISE_POLL_EVERY = point_cloud_utils._PROMISE_POLL_EVERY, 1
  try:
    with pytest.raises(point_cloud_utils.PromiseBroken):
      for _ in range(3):   # (asynchronous: the verdict of call k is examined at call k + 1 or later)
        pano_utils.project_feats_to_equirectangular(good, t(xyz_np), h, w, -1, DEPTH_SCALE)
        torch.cuda.synchronize()
  finally:
    point_cloud_utils._PROMISE_POLL_EVERY = old_every
  # SE3DS_CHECK_PROMISE=1 (module flag _CHECK_PROMISE_SYNC): the debugging mode reads the flag back
  # synchronously behind EVERY promised splat -- the offending call itself raises
  try:
    point_cloud_utils._poll_promise(dev(), force=True)
  except point_cloud_utils.PromiseBroken:
    pass
  old_sync, point_cloud_utils._CHECK_PROMISE_SYNC = point_cloud_utils._CHECK_PROMISE_SYNC, True
  try:
    with pytest.raises(point_cloud_utils.PromiseBroken):
      pano_utils.project_feats_to_equirectangular(good, t(xyz_np), h, w, -1, DEPTH_SCALE)
  finally:
    point_cloud_utils._CHECK_PROMISE_SYNC = old_sync
    try:   # drain the flag this test raised on purpose (it is sticky: later tests share the workspace)
      point_cloud_utils._poll_promise(dev(), force=True)
    except point_cloud_utils.PromiseBroken:
      pass


def test_resize_branches_size_mult_crop_resize_and_mean_padding():
  """The off-path branches of pano_utils that round 2 left as NotImplementedError (VERDICT r2,
  missing #5): equirectangular_to_pointcloud(size_mult != 1) (pano_utils.py:203-208),
  crop_pano(resize_to_original=True) (:299-302) and project_perspective_image(pad_mode='mean')
  (:403-407), each against oracle/warp_np.py (tf.image.resize half-pixel centres: index selection
  and lerp op for op, so nearest AND bilinear are bit-exact)."""
  rng = np.random.default_rng(23)
  n, h, w = 2, 24, 48
  rgb = rng.integers(0, 256, (n, h, w, 3)).astype(np.int32)
  depth = rng.uniform(0, 1, (n, h, w)).astype(F32)
  depth[rng.uniform(size=depth.shape) < 0.05] = 0.0
  for size_mult, method in ((0.5, 'nearest'), (2.0, 'nearest'), (1.5, 'bilinear'), (0.75, 'bilinear')):
    x_o, f_o = warp_np.equirectangular_to_pointcloud(rgb, depth, -1, DEPTH_SCALE, size_mult, method)
    x_g, f_g = pano_utils.equirectangular_to_pointcloud(t(rgb), t(depth), -1, DEPTH_SCALE, size_mult,
                                                        method)
    assert tuple(x_g.shape) == x_o.shape and tuple(f_g.shape) == f_o.shape, (size_mult, method)
    np.testing.assert_array_equal(x_g.cpu().numpy(), x_o, err_msg=str((size_mult, method)))
    np.testing.assert_array_equal(f_g.cpu().numpy(), f_o, err_msg=str((size_mult, method)))
  # plain resizes incl. uint8 and a non-integer factor
  img = rng.integers(0, 256, (1, 10, 14, 2)).astype(np.uint8)
  np.testing.assert_array_equal(pano_utils.resize(t(img), 23, 9, 'nearest').cpu().numpy(),
                                warp_np._resize_nearest(img, 23, 9))
  np.testing.assert_array_equal(pano_utils.resize(t(img), 23, 9, 'bilinear').cpu().numpy(),
                                warp_np._resize_bilinear(img, 23, 9))
  # crop_pano with the resize back, float and integer panos, 3-D and 4-D
  pano = rng.uniform(0, 1, (2, 32, 64, 3)).astype(F32)
  for arr in (pano, pano[0], (pano * 255).astype(np.int32)):
    for method in ('bilinear', 'nearest'):
      want = warp_np.crop_pano(arr, 0.125, method, True)
      got = pano_utils.crop_pano(t(arr), 0.125, method, True).cpu().numpy()
      assert got.shape == arr.shape and got.dtype == arr.dtype
      np.testing.assert_array_equal(got, want)
  with pytest.raises(NotImplementedError):
    pano_utils.crop_pano(t(pano), 0.125, 'bicubic', True)
  # pad_mode='mean': the padding constant is the image mean (binary64 accumulation on both sides)
  img = rng.uniform(0, 1, (24, 24, 3)).astype(F32)
  fov = np.array([np.pi / 2, np.pi / 2], F32)
  p_o = warp_np.project_perspective_image(img, fov, 16, rotations=np.array([0.1, 0.4], F32),
                                          pad_mode='mean')
  p_g = pano_utils.project_perspective_image(t(img), fov, 16, rotations=np.array([0.1, 0.4], F32),
                                             pad_mode='mean').cpu().numpy()
  assert (np.abs(p_g - p_o).max(-1) <= 1e-4).mean() > 0.995
  outside = np.abs(p_o - img.mean()).max(-1) < 1e-6     # pixels that see only the padding
  assert outside.mean() > 0.3 and np.abs(p_g[outside] - np.float32(img.astype(np.float64).mean())).max() < 1e-6


@pytest.mark.parametrize('h', [64, 1024])
def test_device_fast_screen_error_bound(h):
  """The fast index screen as the DEVICE evaluates it (hardware reciprocal / square root inside,
  include/se3ds_geom_math.h): every index it decides equals the exact chain's, and its deviation
  from the exact (fx, fy) stays >= 5x below the margin -- over 4.4 M points incl. the wrap, the
  poles, the equator, denormal-scale clouds and exact ties."""
  rng = np.random.default_rng(50 + h)
  w, m = 2 * h, 400_000
  clouds = [rng.standard_normal((3, m)) * rng.uniform(0.01, 20, (1, m)) for _ in range(5)]
  for axis, scale in [(0, 1e-4), (1, 1e-4), (2, 1e-5)]:
    v = rng.standard_normal((3, m))
    v[axis] *= scale
    clouds.append(v)
  v = rng.standard_normal((3, m))
  v[:2] *= 1e-3
  clouds.append(v)
  clouds.append(rng.standard_normal((3, m)) * 1e-18)
  clouds.append(rng.integers(-3, 4, (3, m)).astype(np.float64))
  margin = 2.0e-6   # SE3DS_FAST_MARGIN
  decided_total = 0
  worst = [0.0, 0.0]
  for c in clouds:
    xyz = c.astype(F32)
    proj = warp_c.equirect_project_coords(np.concatenate([xyz, np.ones((1, m), F32)])[None])[0]
    px, py, pz = proj[0], proj[1], proj[2]
    with np.errstate(divide='ignore', invalid='ignore'):
      vx = np.where(pz == 0, F32(0), px / pz).astype(F32)
      vy = np.where(pz == 0, F32(0), py / pz).astype(F32)
    fx = ((vx + F32(1)) / F32(2) * F32(w)).astype(F32)
    fy = ((vy + F32(1)) / F32(2) * F32(h)).astype(F32)
    ok = (fx > -1) & (fx < w) & (fy > -1) & (fy < h) & (pz > 0)
    idx = np.where(ok, np.trunc(fy).astype(np.int64) * w + np.trunc(fx).astype(np.int64), -1)
    gx, gy, verdict = point_cloud_utils.debug_fast_fxy(t(xyz), h, w)
    gx, gy, verdict = gx.cpu().numpy(), gy.cpu().numpy(), verdict.cpu().numpy()
    dec = verdict >= -1
    decided_total += int(dec.sum())
    np.testing.assert_array_equal(verdict[dec], idx[dec])
    # (rad >= 1e-14: the range the device screen ACCEPTS points in -- below it its unscaled square root and
    #  quotient are not the IEEE ones, and every such point takes the exact chain whatever fx / fy say)
    fin = np.isfinite(fx) & np.isfinite(gx) & np.isfinite(fy) & np.isfinite(gy) & (pz >= 1e-14)
    assert not (dec & (pz < 1e-14)).any()
    dx = np.abs(gx[fin].astype(np.float64) - fx[fin]) / w
    dy = np.abs(gy[fin].astype(np.float64) - fy[fin]) / h
    dx = dx[dx < 0.99]   # the heading wrap: both chains sit on an integer, the screen abstains
    worst = [max(worst[0], dx.max() if dx.size else 0.0), max(worst[1], dy.max() if dy.size else 0.0)]
    assert dx.size == 0 or dx.max() <= margin / 5, dx.max()
    assert dy.size == 0 or dy.max() <= margin / 5, dy.max()
  print(f'device screen, height {h}: max |dfx| / W = {worst[0]:.3g}, max |dfy| / H = {worst[1]:.3g} (margin {margin:g})')
  assert decided_total > 0.5 * len(clouds) * m


@pytest.mark.parametrize('eq_h', [128, 1024])
def test_fused_perspective_paths_match_the_op_chain(eq_h):
  """SURVEY 8f-4 (RE10K notebook cells 15 / 17): the fused perspective -> point cloud and
  equirect -> perspective-guidance kernels are bit-identical to the chain of the individual ops
  (project_perspective_image x2, cast, equirectangular_to_pointcloud; project + 3 x
  get_perspective_from_equirectangular_image + the mask / clip glue), each of which is tested
  against the oracle on its own (test_bilinear_and_resampling_paths); the direct oracle check of
  the fused kernels is test_fused_perspective_paths_vs_oracle_notebook_chain below."""
  from se3ds_amd.models.models import _quantize
  rng = np.random.default_rng(60 + eq_h)
  ph, pw = 64, 64
  k = np.array([[32., 0., 32.], [0., 32., 32.], [0., 0., 1.]], F32)
  rot = np.array([[0.99992853, -0.01185191, 0.00158339], [0.01187317, 0.9998291, -0.01416931],
                  [-0.00141519, 0.01418709, 0.9998984]], F32)
  rgb_u8 = rng.integers(1, 256, (ph, pw, 3)).astype(np.uint8)
  depth = rng.uniform(0.02, 0.2, (ph, pw)).astype(F32)
  rgb = _quantize(t(rgb_u8), torch.float32, mul=float(F32(1.0 / 255.0)), lo=0.0, hi=1.0)
  # ---- cell 15, op by op
  rgb_eq = pano_utils.project_perspective_image(rgb, None, eq_h, camera_intrinsics=k,
                                                rotation_matrix=rot, round_to_nearest=True)
  depth_eq = pano_utils.project_perspective_image(t(depth)[..., None], None, eq_h, camera_intrinsics=k,
                                                  rotation_matrix=rot, round_to_nearest=True)
  feats_eq = _quantize(rgb_eq, torch.int32, mul=255.0, lo=-1e9, hi=1e9)
  xyz_c, f_c = pano_utils.equirectangular_to_pointcloud(feats_eq[None], depth_eq[None, ..., 0], -1, 20.0)
  # ---- fused
  xyz_f, f_f = pano_utils.perspective_to_pointcloud(rgb, t(depth), eq_h, -1, 20.0, camera_intrinsics=k,
                                                    rotation_matrix=rot, round_to_nearest=True)
  assert torch.equal(xyz_c, xyz_f) and torch.equal(f_c, f_f)
  assert int((f_f[0, :, 0] >= 0).sum()) > 0.02 * eq_h * 2 * eq_h   # the frustum covers part of the pano
  # ---- cell 17: move, splat, then back to the perspective view
  rel = t(np.array([[0.0, 0.01, 0.0]], F32))
  ang = 15.0 / 180.0 * np.pi
  new_rot = (rot.astype(np.float64) @ np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0],
                                                [-np.sin(ang), 0, np.cos(ang)]])).astype(F32)
  pred_depth, pred_rgb = pano_utils.project_feats_to_equirectangular(f_f, xyz_f, eq_h, 2 * eq_h, -1,
                                                                     20.0, offset=rel)
  g_rgb = pano_utils.get_perspective_from_equirectangular_image(pred_rgb[0], k, new_rot, ph, pw)
  # (tensor / tensor: IEEE division as tf.truediv; torch turns `x / 255` into x * (1 / 255))
  g_rgb = torch.clamp(g_rgb / torch.full_like(g_rgb, 255.0), 0, 1)[None]
  g_d = pano_utils.get_perspective_from_equirectangular_image(pred_depth[0][..., None], k, new_rot, ph, pw)[None]
  m_eq = ((pred_depth != 1.0) & (pred_depth != 0.0) & torch.all(pred_rgb != 0.0, dim=-1)).float()
  g_m = pano_utils.get_perspective_from_equirectangular_image(m_eq[0][..., None], k, new_rot, ph, pw)
  pm = (g_m[None] == 1.0).float()
  pi_f, pd_f, pm_f = pano_utils.perspective_guidance(pred_rgb[0], pred_depth[0], k, new_rot, ph, pw)
  assert torch.equal(pm_f, pm) and torch.equal(pd_f, pm * g_d) and torch.equal(pi_f, pm * g_rgb)
  assert 0.05 < float(pm_f.mean()) <= 1.0


@pytest.mark.parametrize('eq_h', [128, 512])
def test_fused_perspective_paths_vs_oracle_notebook_chain(eq_h):
  """the fused kernels against the ORACLE's composition of RE10K notebook cells 15 and
  17 (oracle/warp_np.notebook_cell15_pointcloud / notebook_cell17_guidance, which chain the
  restatements of pano_utils.py:344-476 and :164-242).  The perspective paths are not bit-exact
  against NumPy (3x3 fp32 products are not pinned to one summation order, see
  test_bilinear_and_resampling_paths): cell 15 rounds its sampling coordinates to integers, so
  points are IDENTICAL except where a coordinate sits within an ulp of a half-integer; cell 17's
  bilinear gathers agree to 1e-4 except on the heading seam.  Both bars: > 99.5 % of the elements."""
  rng = np.random.default_rng(70 + eq_h)
  ph, pw = 64, 64
  k = np.array([[32., 0., 32.], [0., 32., 32.], [0., 0., 1.]], F32)
  rot = np.array([[0.99992853, -0.01185191, 0.00158339], [0.01187317, 0.9998291, -0.01416931],
                  [-0.00141519, 0.01418709, 0.9998984]], F32)
  rgb01 = (rng.integers(1, 256, (ph, pw, 3)).astype(F32) / F32(255)).astype(F32)
  depth = rng.uniform(0.02, 0.2, (ph, pw)).astype(F32)
  # ---- cell 15
  xyz_o, f_o = warp_np.notebook_cell15_pointcloud(rgb01, depth, k, rot, eq_h)
  xyz_g, f_g = pano_utils.perspective_to_pointcloud(t(rgb01), t(depth), eq_h, -1, 20.0,
                                                    camera_intrinsics=k, rotation_matrix=rot,
                                                    round_to_nearest=True)
  xyz_g, f_g = xyz_g.cpu().numpy(), f_g.cpu().numpy()
  assert xyz_g.shape == xyz_o.shape and f_g.shape == f_o.shape and f_g.dtype == f_o.dtype
  same = np.all(xyz_g == xyz_o, axis=1)[0] & np.all(f_g == f_o, axis=-1)[0]
  covered = f_o[0, :, 0] >= 0
  assert covered.mean() > 0.02, covered.mean()             # the frustum covers part of the panorama
  assert same.mean() > 0.995 and same[covered].mean() > 0.995, (same.mean(), same[covered].mean())
  # ---- cell 17: splat the ORACLE's cloud with the oracle (one input for both guidance paths)
  rel = np.array([[0.0, 0.01, 0.0]], F32)
  ang = 15.0 / 180.0 * np.pi
  new_rot = (rot.astype(np.float64) @ np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0],
                                                [-np.sin(ang), 0, np.cos(ang)]])).astype(F32)
  d_o, r_o = warp_c.project_feats_to_equirectangular(f_o, xyz_o, eq_h, 2 * eq_h, -1, 20.0, offset=rel)
  want = warp_np.notebook_cell17_guidance(r_o[0], d_o[0], k, new_rot, ph, pw)
  got = pano_utils.perspective_guidance(t(r_o[0]), t(d_o[0]), k, new_rot, ph, pw)
  assert 0.05 < float(want[2].mean()) <= 1.0
  for name, a, b in zip(('proj_image', 'proj_depth', 'proj_mask'), got, want):
    a = a.cpu().numpy()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    close = np.abs(a - b).max(-1) <= 1e-4
    assert close.mean() > 0.995, (name, close.mean())
