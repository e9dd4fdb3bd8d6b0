"""Where the two sorted-splat kernels spend their time, per phase (a -DSE3DS_PROBE build of geom.hip,
tools/probes/warp_phases.sh; never the shipped library): every workgroup stamps the 100 MHz wall
clock at its phase boundaries; this prints, per kernel, the launch ramp (first to last workgroup
start), each phase's mean / max duration over the workgroups, and the kernel's span.

  SE3DS_LIB=/tmp/libgeomprobe.so python tools/warp_phases.py [--height 512] [--depth random|room]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from se3ds_amd import _lib as lib_mod  # noqa: E402
from se3ds_amd.utils import point_cloud_utils  # noqa: E402

S1 = ['screen (loads, fp32 screen, histogram)', 'queue the undecided', 'binary64 pass + barrier',
      'scan + run descriptors', 'placement in LDS', 'chunk stores issued', 'sink partials']
S2 = ['run row -> compact prefix', 'sink z (first) + LDS tile init', 'pass A: search, gather, z-min',
      'pass B: features / occluded', 'sink slots', 'outputs stored', 'first: fold partials; ticket']


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--height', type=int, default=512)
  ap.add_argument('--depth', default='random')
  ap.add_argument('--views', type=int, default=2)
  args = ap.parse_args()
  dev = torch.device('cuda:0')
  h, w = args.height, 2 * args.height
  rng = np.random.default_rng(1234)
  panos, target = bench._warp_inputs(rng, h, w, args.views, dev, args.depth)
  g = [(torch.from_numpy(r).to(dev), torch.from_numpy(d).to(dev), torch.from_numpy(p).to(dev))
       for r, d, p in panos]
  tgt = torch.from_numpy(target).to(dev)
  M = args.views * h * w
  mem = point_cloud_utils.PointCloudMemory(1, 3, torch.int32, dev, capacity=M)
  mem.append_views_and_project(g, -1, 20.0, tgt, h, w, with_mask=True)
  L = lib_mod.lib()
  if not hasattr(L, 'se3ds_geom_probe_read'):
    raise SystemExit('not a -DSE3DS_PROBE build: run tools/probes/warp_phases.sh')
  L.se3ds_geom_probe_read.restype = ctypes.c_int
  L.se3ds_geom_probe_read.argtypes = [ctypes.c_void_p]
  d_o = torch.empty((1, h, w), dtype=torch.float32, device=dev)
  f_o = torch.empty((1, h, w, 3), dtype=torch.float32, device=dev)
  m_o = torch.empty((1, h, w), dtype=torch.float32, device=dev)
  ws = point_cloud_utils._workspace(L.se3ds_splat_workspace_bytes(1, M, h, w, 3), dev)
  hint = point_cloud_utils.FEAT_BYTE_RANGE if mem.byte_range else 0
  st = lib_mod.stream()
  def project_c():
    lib_mod.check(L.se3ds_project_equirect_memory(
        mem._x.data_ptr(), tgt.data_ptr(), mem._f.data_ptr(), lib_mod.I32 | hint, 1, M, mem.capacity, 3,
        h, w, 20.0, -1.0, 0.0, d_o.data_ptr(), f_o.data_ptr(), m_o.data_ptr(), -1.0, ws.data_ptr(),
        ws.numel(), st), 'se3ds_project_equirect_memory')
  buf = np.zeros((2, 4096, 12), np.uint64)
  for rep in range(3):   # (the last of a few back-to-back calls: warm caches, steady clocks)
    for _ in range(20):
      project_c()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(50):
      project_c()
    ev1.record()
    torch.cuda.synchronize()
  print(f'{h}x{w} V={args.views} {args.depth}: {1e3 * ev0.elapsed_time(ev1) / 50:.1f} us per render (probe build)')
  assert L.se3ds_geom_probe_read(buf.ctypes.data) == 0
  for k, names in ((0, S1), (1, S2)):
    t = buf[k].astype(np.int64)
    used = t[:, 0] != 0
    t = t[used] * 10.0 / 1e3   # 100 MHz ticks -> us
    n = t.shape[0]
    t0 = t[:, 0].min()
    print(f'--- kernel {"S1 splat_sort" if k == 0 else "S2 splat_sort_resolve"}: {n} workgroups, '
          f'span {t[:, 7].max() - t0:.2f} us, starts spread over {t[:, 0].max() - t0:.2f} us, '
          f'workgroup lifetime mean {np.mean(t[:, 7] - t[:, 0]):.2f} / max {np.max(t[:, 7] - t[:, 0]):.2f} us')
    for i, name in enumerate(names):
      d = t[:, i + 1] - t[:, i]
      print(f'  {name:42s} mean {d.mean():6.2f}  max {d.max():6.2f} us')
  if buf[0][:, 0].any() and buf[1][:, 0].any():
    a, b = buf[0].astype(np.int64), buf[1].astype(np.int64)
    gap = (b[b[:, 0] != 0][:, 0].min() - a[a[:, 0] != 0][:, 7].max()) * 10.0 / 1e3
    print(f'last S1 workgroup end -> first S2 workgroup start: {gap:.2f} us')


if __name__ == '__main__':
  main()
