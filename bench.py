"""bench.py -- SE3DS hot-path throughput on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload gan_step|warp]

Prints ONE JSON line on rank 0 (contract in the task statement).  With --gpus N > 1 and no
torchrun environment, this process only LAUNCHES `python -m torch.distributed.run
--nproc-per-node N bench.py ...` as a child (before anything touches the GPU) and relays its
output: one rank per GPU over RCCL, as the reference picks its strategy from the device count
(main.py:55-63).  Under torchrun, WORLD_SIZE must equal --gpus.  Workloads:
  gan_step  (default) one G+D train step (train_g_d) at 512x1024 RGB-D, bf16 compute,
            random-init ResNet-101 G / multi-scale SN-PatchGAN D, synthetic panoramas.
            metric = panoramas/sec.  Data-parallel over ranks (weak scaling).
  warp      1024x2048 equirect: 2 source views unprojected, 1 target rendered per step
            (SURVEY 8d, cfg5).  Replicas only.
The CPU baseline leg times the oracle (oracle/, a restatement -- TF cannot run here) on a
bounded sample on the host cores; it is a reported baseline, never the thing measured.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BF16_PEAK_TFLOPS = 2500.0  # dense MFMA bf16
F32_PEAK_TFLOPS = 157.3


def _dist_setup(ngpus):
  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local = int(os.environ.get('LOCAL_RANK', '0'))
  # test hook: SE3DS_BENCH_BACKEND=gloo runs several ranks on ONE GPU (RCCL refuses that), which
  # exercises this file's multi-rank plumbing on a 1-GPU box
  backend = os.environ.get('SE3DS_BENCH_BACKEND', 'nccl')
  if backend != 'nccl':
    local = 0
  if backend == 'nccl' and torch.cuda.device_count() < max(world, ngpus):
    # fail fast, before any rendezvous can hang.  (device_count() may initialise the HIP runtime
    # on builds without amdsmi: harmless here, this process is a rank already -- but the launcher
    # below must stay spawn-only, never re-exec, for exactly that reason)
    raise SystemExit(f'bench.py: --gpus {ngpus} (WORLD_SIZE {world}) needs one GPU per rank, this '
                     f'node has {torch.cuda.device_count()}')
  torch.cuda.set_device(local)
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    # single-node job: keep the bootstrap / rendezvous sockets on loopback (the hostname of the
    # GPU boxes does not resolve; c10d logs it on every run)
    os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
    os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
    if backend == 'nccl':
      dist.init_process_group('nccl', rank=rank, world_size=world,
                              device_id=torch.device('cuda', local))
    else:
      dist.init_process_group(backend, rank=rank, world_size=world)
  return rank, world, local


def _launch_ranks(n):
  """--gpus N without a torchrun environment: start N ranks as a CHILD process (never exec: this
  must also work from a process that has initialised the GPU) and relay its output."""
  import socket
  import subprocess
  if os.environ.get('SE3DS_BENCH_BACKEND', 'nccl') == 'nccl' and torch.cuda.device_count() < n:
    sys.stderr.write(f'bench.py: --gpus {n} needs {n} GPUs on this node, found '
                     f'{torch.cuda.device_count()} (one process per GPU over RCCL)\n')
    return 2
  with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
         '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + \
      sys.argv[1:]
  return subprocess.run(cmd, env=env).returncode


from se3ds_amd.bench_util import pmc_traffic as _pmc_traffic  # noqa: E402


def _collective_preflight(rank, world, dev):
  """Before the timed loop of a multi-rank run: which ranks the communicator really sees, the bus
  bandwidth of one 256 MiB all-reduce (the gradient buckets' size class) and the latency of one
  [2][1024] fp32 all-reduce (a SyncBatchNormalization statistics exchange, ~390 per step) -- so that
  the first scaling curve explains itself: ring all-reduce over xGMI is per-link bound (7 links x
  ~153 GB/s per GPU), small collectives are latency bound."""
  out = {}
  try:
    ids = [None] * world
    dist.all_gather_object(ids, (rank, torch.cuda.current_device() if dev.type == 'cuda' else -1))
    out['rccl_ranks_seen'] = sorted(i[0] for i in ids)
    out['devices'] = [i[1] for i in sorted(ids)]
    # 256 MiB over RCCL (gloo -- the one-GPU plumbing tests -- goes through the host: 16 MiB)
    nbig = (64 if dist.get_backend() == 'nccl' else 4) * 1024 * 1024
    big = torch.ones(nbig, dtype=torch.float32, device=dev)
    small = torch.ones((2, 1024), dtype=torch.float32, device=dev)
    def timed(t, reps):
      dist.all_reduce(t)
      torch.cuda.synchronize()
      dist.barrier()
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(reps):
        dist.all_reduce(t)
      torch.cuda.synchronize()
      return (time.perf_counter() - t0) / reps
    tb = _max_over_ranks(timed(big, 5), world, dev)
    ts = _max_over_ranks(timed(small, 50), world, dev)
    nbytes = big.numel() * 4
    out['allreduce_big_bytes'] = nbytes
    out['allreduce_big_ms'] = 1e3 * tb
    out['allreduce_big_algbw_GBs'] = nbytes / tb / 1e9
    out['allreduce_big_busbw_GBs'] = nbytes / tb / 1e9 * 2 * (world - 1) / world
    out['allreduce_syncbn_2x1024_us'] = 1e6 * ts
    out['sum_check'] = float(big[0].item())   # (world ** reps: the collective really summed)
    del big, small
    torch.cuda.empty_cache()
  except Exception as e:   # noqa: BLE001  (a diagnostic must not take the bench down)
    out['error'] = repr(e)[:300]
  return out


def _barrier(world):
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()


def _max_over_ranks(x, world, dev):
  if world == 1:
    return x
  t = torch.tensor([x], dtype=torch.float64, device=dev)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())


# ----------------------------------------------------------------------------- warp workload
def _room_depth(rng, h, w, pos):
  """Depth of a 9 x 7 x 3 m box room seen from `pos` (metres / 20), half-pixel-centre equirect
  grid as pano_utils.py:211-218, plus 1 cm noise: a smooth map like real scans."""
  hp = 0.5 * np.pi / h
  el = np.linspace(hp, np.pi - hp, h)[:, None]
  hd = np.linspace(1.5 * np.pi - hp, -0.5 * np.pi + hp, w)[None, :]
  d = np.stack([np.sin(el) * np.cos(hd), np.sin(el) * np.sin(hd), np.cos(el) * np.ones_like(hd)])
  lo = np.array([-4.5, -3.5, -1.5])[:, None, None] - pos.reshape(3, 1, 1)
  hi = np.array([4.5, 3.5, 1.5])[:, None, None] - pos.reshape(3, 1, 1)
  with np.errstate(divide='ignore', invalid='ignore'):
    t = np.where(d > 0, hi / d, np.where(d < 0, lo / d, np.inf))
  dist = t.min(0) + rng.normal(0, 0.01, (h, w))
  return (dist / 20.0).astype(np.float32)[None]


def _warp_inputs(rng, h, w, views, dev, depth_kind='random'):
  panos = []
  for _ in range(views):
    rgb = rng.integers(0, 256, (1, h, w, 3)).astype(np.int32)
    depth = rng.uniform(0, 1, (1, h, w)).astype(np.float32)
    if depth_kind == 'room':
      pos = (rng.standard_normal((1, 3)) * 0.5).astype(np.float32)
      depth = _room_depth(rng, h, w, pos[0].astype(np.float64))
      poison = rng.uniform(0, 1, (1, h, w))
      depth[poison < 0.02] = 0.0
      panos.append((rgb, depth, pos))
      continue
    poison = rng.uniform(0, 1, (1, h, w))
    depth[poison < 0.02] = 0.0
    depth[poison > 0.99] = 1.0
    pos = (rng.standard_normal((1, 3)) * 0.5).astype(np.float32)
    panos.append((rgb, depth, pos))
  target = (rng.standard_normal((1, 3)) * 0.5).astype(np.float32)
  return panos, target


def bench_warp(args, rank, world, dev):
  from se3ds_amd.utils import pano_utils, point_cloud_utils
  h, w, views = args.warp_height, 2 * args.warp_height, 2
  rng = np.random.default_rng(1234 + rank)
  panos, target = _warp_inputs(rng, h, w, views, dev, args.warp_depth)
  g = [(torch.from_numpy(r).to(dev), torch.from_numpy(d).to(dev), torch.from_numpy(p).to(dev))
       for r, d, p in panos]
  tgt = torch.from_numpy(target).to(dev)
  P = h * w

  M = views * P
  # the point-cloud memory of the trajectory loops (eval_metric.py:144-239), preallocated
  mem = point_cloud_utils.PointCloudMemory(1, 3, torch.int32, dev, capacity=M)
  frame = (torch.empty((1, h, w), dtype=torch.float32, device=dev),
           torch.empty((1, h, w, 3), dtype=torch.float32, device=dev),
           torch.empty((1, h, w), dtype=torch.float32, device=dev))

  def step():
    # ONE library call per trajectory step (se3ds_warp_views_to_target, round 5): every view is
    # unprojected straight into its window of the memory (the concat of eval_metric.py:238-239 /
    # models.py:239-245 without a copy), then one target is rendered (the four separate calls of round 4
    # took 154.5 us per step against 117).
    mem.clear()
    # (the frame buffers are reused, as a trajectory loop does)
    return mem.append_views_and_project(g, -1, 20.0, tgt, h, w, with_mask=True, out=frame)

  for _ in range(args.warmup):
    step()
  _barrier(world)
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step()
  _barrier(world)
  dt = _max_over_ranks(time.perf_counter() - t0, world, dev)

  # dominant kernel chain = project+splat on a resident memory: HIP events on the launch stream
  step()
  def timed(fn, reps):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
      fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps
  # The kernels are timed through the C ABI with preallocated outputs (a ctypes call costs ~3-5 us of
  # host time; the Python wrappers -- output allocation, workspace and byte-range bookkeeping -- cost
  # 15-40 us per call, more than the 512 x 1024 kernels take, and would be what is measured).
  from se3ds_amd import _lib as lib_mod
  from se3ds_amd.utils import _host_tables
  L = lib_mod.lib()
  reps = max(20, args.steps)
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
  proj_ms = timed(project_c, reps)
  algo_bytes = 28 * M + 20 * P  # 28 B/point in, 16 B/px out + 4 B/px mask
  achieved = algo_bytes / (proj_ms * 1e-3) / 1e9
  # the other kernel of the step: one view's unproject (44 B per pixel algorithmic: 4 depth + 12 features in,
  # 16 coordinates + 12 features out; into a point-cloud memory, whose homogeneous row is preset, it moves 40),
  # into its window of the memory
  rgb0, depth0, pos0 = g[0]
  tab = _host_tables.equirect_tables(h, w, dev)
  tb = tab.data_ptr()
  def unproject_c():
    lib_mod.check(L.se3ds_unproject_equirect_into(
        rgb0.data_ptr(), lib_mod.I32 | point_cloud_utils.XYZ1_ONES_PRESET, depth0.data_ptr(), tb, tb + 4 * h,
        tb + 8 * h, tb + 8 * h + 4 * w,
        pos0.data_ptr(), 1, h, w, 3, -1.0, 20.0, mem._x.data_ptr(), mem._f.data_ptr(), mem.capacity, 0,
        st), 'se3ds_unproject_equirect_into')
  unp_alone_ms = timed(unproject_c, reps)
  unp_bytes = 44 * P
  # ... and IN THE PIPELINE (re-launched alone on reused buffers the kernel writes
  # 59 MB into a 256 MB MALL that still holds them and round 5's line read 0.71 of peak where the
  # profile of the step says 0.58).  Device time of the whole step = the same one-call entry point the
  # wrapper uses (V unprojects + S1 + S2, HIP events around `reps` back-to-back ctypes calls), minus the
  # project + splat chain timed the same way, over the V views: every unproject writes its window
  # behind another view's, the splat streams both, nothing is re-used between kernels.
  import ctypes
  arr = lambda xs: (ctypes.c_void_p * views)(*xs)
  vf, vd, vp = arr([x[0].data_ptr() for x in g]), arr([x[1].data_ptr() for x in g]), arr([x[2].data_ptr() for x in g])
  def step_c():
    lib_mod.check(L.se3ds_warp_views_to_target(
        vf, lib_mod.I32 | hint | point_cloud_utils.XYZ1_ONES_PRESET, vd, vp, views, 1, h, w, 3, -1.0, 20.0, tb,
        tb + 4 * h, tb + 8 * h,
        tb + 8 * h + 4 * w, mem._x.data_ptr(), mem._f.data_ptr(), mem.capacity, 0, tgt.data_ptr(), h, w, 0.0,
        d_o.data_ptr(), f_o.data_ptr(), m_o.data_ptr(), -1.0, ws.data_ptr(), ws.numel(), st),
        'se3ds_warp_views_to_target')
  dev_step_ms = timed(step_c, reps)
  unp_ms = max(dev_step_ms - proj_ms, 0.0) / views

  traffic, traffic_detail = (None, None)
  if (h, views, args.warp_depth) == (1024, 2, 'random'):
    traffic, traffic_detail = _pmc_traffic('r06_warp_pmc.json', ('splat_sort',))
  out = {
      'metric': 'panoramas/sec (2-view unproject + 1 target render, 1024x2048 equirect)',
      'value': world * args.steps / dt, 'unit': 'panoramas/sec', 'n_gpus': world,
      'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
      'data': 'synthetic',
      'config': {'workload': f'warp cfg5 {h}x{w} V={views} (replicas only)' +
                             ('' if args.warp_depth == 'random' else f', {args.warp_depth} depth')},
      'roofline': {'bound': 'hbm', 'kernel': 'project+splat (splat_sort_kernel: pack + sort each chunk by target row; '
                             'splat_sort_resolve_kernel: gather, z-min, features, outputs, sink)',
                   'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                   'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                   'traffic_detail': traffic_detail,
                   'ms_per_launch': proj_ms, 'algorithmic_bytes': algo_bytes,
                   # VALU side of the two splat kernels (tracked PMC pass, profiles/r06_warp_valu_pmc.json)
                   'valu_frac': _warp_valu_frac() if (h, views) == (1024, 2) else None,
                   'unproject': {'kernel': 'unproject_equirect_vec4_kernel (one view)',
                                 'how': '(device time of the step - project + splat) / views: the kernel as it '
                                        'runs in the pipeline, launch gap included',
                                 'ms_per_launch': unp_ms, 'algorithmic_bytes': unp_bytes,
                                 'achieved': unp_bytes / (unp_ms * 1e-3) / 1e9, 'unit': 'GB/s',
                                 'frac': unp_bytes / (unp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 'alone_ms_per_launch': unp_alone_ms,
                                 'alone_frac': unp_bytes / (unp_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                   # device time of one step (events around back-to-back C calls of the one-call entry
                   # point) and the host's share: wall per step through the Python API over that
                   'device_ms_per_step': dev_step_ms,
                   'step_over_kernels': (1e3 * dt / args.steps) / dev_step_ms},
  }
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    out['cpu_baseline'] = cpu_baseline_warp(panos, target, h, w)
  return out


def _warp_valu_frac():
  """VALU-busy fraction of the splat kernels from the tracked counter pass, or None."""
  path = os.path.join(ROOT, 'profiles', 'r06_warp_valu_pmc.json')
  try:
    with open(path) as f:
      d = json.load(f)
    return {k: v.get('valu_busy_frac') for k, v in d.get('kernels', {}).items()}
  except (OSError, ValueError):
    return None


def cpu_baseline_warp(panos, target, h, w):
  """Oracle (C restatement, 1 thread) on the same inputs, bounded to a few runs."""
  from oracle import warp_c, warp_np
  t0 = time.perf_counter()
  runs = 0
  while runs < 2 and time.perf_counter() - t0 < 30:
    xs, fs = [], []
    tabs = warp_np.equirect_angle_tables(h, w)
    for rgb, depth, pos in panos:
      x, f = warp_c.unproject_equirect(rgb, depth, tabs, -1, 20.0, position=pos)
      xs.append(x)
      fs.append(f)
    warp_c.project_feats_to_equirectangular(np.concatenate(fs, 1), np.concatenate(xs, 2), h, w,
                                            -1, 20.0, offset=target)
    runs += 1
  dt = (time.perf_counter() - t0) / runs
  return {'value': 1.0 / dt, 'unit': 'panoramas/sec', 'cores': 1, 'kind': 'port',
          'sample': f'{runs} full warp steps at {h}x{w} (C oracle restatement, not TF)'}


# ------------------------------------------------------------------------- gan_step workload
def bench_gan_step(args, rank, world, dev):
  from se3ds_amd import bench_step
  return bench_step.run(args, rank, world, dev, _barrier, _max_over_ranks)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--workload', default=None, choices=['gan_step', 'warp'])
  ap.add_argument('--warp-height', type=int, default=1024)
  ap.add_argument('--warp-depth', default='random', choices=['random', 'room'],
                  help='random: independent depth per pixel (worst case for the splat, default); '
                       'room: smooth box-room depth like a real scan')
  ap.add_argument('--batch', type=int, default=0, help='per-GPU batch for gan_step (0 = auto)')
  ap.add_argument('--image-size', type=int, default=512)
  ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-batch-max', action='store_true',
                  help='gan_step: skip the extra large-batch measurement (batch_max) on the default line')
  ap.add_argument('--no-shipped', action='store_true',
                  help='gan_step: skip the extra d_step_per_g_step = 2 cluster-step measurement')
  ap.add_argument('--no-fp32', action='store_true',
                  help='gan_step: skip the extra fp32 (reference arithmetic) step measurement')
  ap.add_argument('--no-warp', action='store_true',
                  help='gan_step: skip the extra cfg5 warp measurement on the default line')
  args = ap.parse_args()
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    sys.exit(_launch_ranks(args.gpus))
  rank, world, local = _dist_setup(args.gpus)
  if world != args.gpus:
    raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with '
                     f'--nproc-per-node {args.gpus} (or without torchrun: bench.py spawns the ranks)')
  dev = torch.device('cuda', local)
  preflight = _collective_preflight(rank, world, dev) if world > 1 else None
  workload = args.workload
  if workload is None:
    workload = 'gan_step' if os.path.exists(os.path.join(ROOT, 'se3ds_amd', 'bench_step.py')) \
        else 'warp'
  out = bench_warp(args, rank, world, dev) if workload == 'warp' else \
      bench_gan_step(args, rank, world, dev)
  if workload == 'gan_step' and world == 1 and not args.no_warp:
    # the second half of the hot path, driver-observed on the default line: cfg5 warp
    # (1024x2048, 2 views) throughput and the HBM roofline of its project+splat chain
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    wargs = argparse.Namespace(**vars(args))
    wargs.steps, wargs.warmup, wargs.no_cpu_baseline = 50, 5, True
    w = bench_warp(wargs, rank, world, dev)
    out['warp'] = {k: w[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'config', 'roofline')}
    # ... and at north_star's stated size, 512 x 1024 (2 views -> 1 target)
    wargs.warp_height = 512
    w5 = bench_warp(wargs, rank, world, dev)
    r5 = w5['roofline']
    out['warp']['at_512'] = {
        'config': w5['config'], 'value': w5['value'], 'ms_per_step': w5['ms_per_step'],
        'ms_per_launch': r5['ms_per_launch'], 'achieved': r5['achieved'], 'frac': r5['frac'],
        'unit': 'GB/s', 'algorithmic_bytes': r5['algorithmic_bytes'], 'unproject': r5['unproject'],
        'device_ms_per_step': r5['device_ms_per_step'], 'step_over_kernels': r5['step_over_kernels']}
  if preflight is not None:
    out['collectives'] = preflight
  out['backend'] = (dist.get_backend() if world > 1 else None)
  out['world_size'] = (dist.get_world_size() if world > 1 else 1)
  if rank == 0:
    print(json.dumps(out))
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
