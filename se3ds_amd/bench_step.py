"""gan_step workload of bench.py: one `train_g_d` (G + D update, hinge GAN + depth + world
consistency losses, per-tensor clip, Adam, EMA) per step on synthetic 512x1024 RGB-D panoramas
with the shipped highres configuration (ResNet-101 generator, gen_dims 128; 2-scale 6-layer
SN-PatchGAN), random-init weights, bf16 compute / fp32 master weights.  Data-parallel over
ranks: every rank steps its own batch, gradients are summed over RCCL (weak scaling)."""
import os
import time

import torch

from se3ds_amd import gin_lite
from se3ds_amd.hipops import nn
from se3ds_amd.models import image_models
from se3ds_amd.trainers import gan_manager
from se3ds_amd.trainers import se3ds_trainer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA
F32_PEAK_TFLOPS = 157.3


def synth_batch(n, h, seed, device):
  """SURVEY.md section 8d synthetic inputs, generated on the device."""
  g = torch.Generator(device=device).manual_seed(seed)
  w = 2 * h
  r = lambda *s: torch.rand(s, generator=g, device=device)
  image = r(n, h, w, 3)
  depth = r(n, h, w, 1)
  poison = r(n, h, w, 1)
  depth = torch.where(poison < 0.02, torch.zeros_like(depth), depth)
  depth = torch.where(poison > 0.99, torch.ones_like(depth), depth)
  pm = (r(n, h, w, 1) < 0.5).float()
  pm[:, h // 3:h // 3 + max(1, h // 8)] = 0
  bm = torch.zeros((n, h, w, 1), device=device)
  bm[:, :h // 8] = 1
  bm[:, -(h // 8):] = 1
  return dict(image=image, depth=depth, proj_mask=pm, proj_image=image * pm,
              proj_depth=depth * pm, blurred_mask=bm)


def build_gan(args, dev, world):
  gin_lite.clear_config()
  gin_lite.parse_config_files_and_bindings(
      [os.path.join(ROOT, 'configs', 'highres', 'highres.gin')],
      [f'GANManager.image_size = {args.image_size}', 'GANManager.d_step_per_g_step = 1',
       'GANManager.num_batched_steps = 1'] + list(getattr(args, 'gin_bindings', []) or []))
  strategy = gan_manager.DataParallelStrategy(dev) if world > 1 else \
      gan_manager.OneDeviceStrategy(dev)
  dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
  gan = se3ds_trainer.GAN(strategy=strategy, model_dir='', compute_dtype=dtype)
  gan.device_init = True
  gan._create_obj()
  return gan


def run(args, rank, world, dev, barrier, max_over_ranks):
  gan = build_gan(args, dev, world)
  n = args.batch if args.batch > 0 else 8
  h = args.image_size
  batch = synth_batch(n, h, 1234 + rank, dev)
  def step():
    # one cluster step of the host loop (gan_manager.py:409-421 with num_batched_steps = 1):
    # train_g_d, then the host advances global_step.  From the second step on the EMA model
    # is updated by decay and the step-0-only EMA forward (se3ds_trainer.py:258-259) is gone.
    gan.train_g_d(batch)
    gan.global_step += gan.num_batched_steps
  for _ in range(args.warmup):
    step()
  # a full (generation-2) garbage collection of the Python heap costs ~40 ms of host time about
  # every fifth step (tools/step_times.py); run it now so that a short timed region measures the
  # steady state rather than the position of that pause
  import gc
  gc.collect()
  barrier(world)
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step()
  torch.cuda.synchronize()
  own_dt = time.perf_counter() - t0   # (this rank's own time; the barrier below waits for the slowest)
  barrier(world)
  dt = max_over_ranks(time.perf_counter() - t0, world, dev)
  ms = 1e3 * dt / args.steps

  # ---- roofline of the dominant kernel family (implicit-GEMM convolutions): one extra,
  # instrumented step with HIP events around every conv launch on the launch stream
  # (two instrumented steps, the second one counts: the serial path launches kernels the timed
  # schedule never does -- per-layer split reductions, single-layer operand staging -- and their
  # first launch in a process loads code: 0.366-0.372 on a first pass against 0.385-0.393)
  for _ in range(2):
    prof = nn.ConvProfiler()
    nn.set_conv_profiler(prof)
    step()
    torch.cuda.synchronize()
    nn.set_conv_profiler(None)
  summ = prof.summary()
  if os.environ.get('SE3DS_BENCH_SHAPES'):
    # per-shape table of the instrumented step (SE3DS_BENCH_SHAPES=all: every shape, else the top 40)
    rows = sorted(prof.by_shape().items(), key=lambda kv: -kv[1][0])
    if os.environ['SE3DS_BENCH_SHAPES'] != 'all':
      rows = rows[:40]
    for (kind, tag), (ms_, fl, cnt) in rows:
      print(f'SHAPE {ms_:8.2f} ms {fl / max(ms_, 1e-9) / 1e9:7.0f} TF/s x{cnt:3d} {kind:6s} {tag}')
  peak = BF16_PEAK_TFLOPS if args.dtype == 'bf16' else F32_PEAK_TFLOPS
  achieved = summ['flops'] / (summ['ms'] * 1e-3) / 1e12 if summ['ms'] > 0 else 0.0
  m = gan._save_metrics_to_dict()
  # HBM traffic of the dominant kernel (3x3 1024->1024 @32x64 forward, 35 % of the conv time with
  # its data-gradient twin) from the tracked PMC summary; the family is MFMA-bound, the figure
  # shows that no kernel re-reads its operands from HBM (algorithmic: 52 MB in, 33.5 MB out)
  traffic = traffic_detail = None
  if h == 512 and n == 8 and args.dtype == 'bf16':
    from se3ds_amd import bench_util
    traffic, traffic_detail = bench_util.pmc_traffic(
        'r06_conv_pmc_1024_1024_3_1_32_64_1_8.json', ('igemm_halo_kernel<0, 256, 2',))   # (prefix: the kernel has further template arguments)
    if traffic_detail is not None:
      traffic_detail['kernel'] = 'igemm_halo_kernel<0, 256, 2> (3x3 1024->1024 @32x64, batch 8)'
      traffic_detail['algorithmic_mb'] = 52.4 + 33.6
  out = {
      'metric': 'panoramas/sec (G+D train step) at 512x1024 RGB-D' if h == 512 else
                f'panoramas/sec (G+D train step) at {h}x{2 * h} RGB-D',
      'value': world * n * args.steps / dt, 'unit': 'panoramas/sec', 'n_gpus': world,
      'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': True,
      'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
      'config': {'workload': f'configs/highres {h}x{2 * h} full GAN step train_g_d (G+D+hinge+'
                             f'depth+wc losses, clip, Adam, EMA), ResNet-101 gen_dims 128, '
                             f'per-GPU batch {n}, d_step_per_g_step 1',
                 'per_gpu_batch': n, 'global_batch': n * world, 'parallelism': f'dp{world}'},
      'roofline': {'bound': 'mfma', 'kernel': 'implicit-GEMM convolutions: igemm_halo / igemm_big / '
                   'igemm_glds + wgrad_taps / wgrad_glds (every conv fwd, dgrad and wgrad call of '
                   'one step)', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                   'frac': achieved / peak,
                   # `frac`: every conv launch timed ALONE (serial instrumented step, HIP events);
                   # `frac_in_step`: the same FLOPs over the WHOLE timed step (norms, optimiser,
                   # launch gaps and stream sharing included) -- the effective MFMA fraction
                   'frac_in_step': summ['flops'] / (ms * 1e-3) / 1e12 / peak,
                   'traffic': traffic, 'traffic_detail': traffic_detail,
                   'launches': summ['launches'],
                   'avg_launch_ms': summ['ms'] / max(summ['launches'], 1),
                   'conv_ms_per_step': summ['ms'], 'conv_tflop_per_step': summ['flops'] / 1e12,
                   'by_kind': summ['by_kind']},
      'losses': {k: float(v) for k, v in m.items() if k in (
          'gen/gen_gan_loss', 'dis/disc_loss', 'gen/depth_loss', 'gen/wc_loss')},
      'hbm_gib_peak': torch.cuda.max_memory_allocated(dev) / 2**30,
  }
  if world > 1:
    # one more step with the collective log on: what a rank sends per step (the scaling curve's
    # explanation: ~390 latency-bound SyncBN sums + 4.46 GB of gradient buckets per step)
    from se3ds_amd.trainers import dist_utils
    dist_utils.COLLECTIVE_LOG = []
    step()
    torch.cuda.synchronize()
    out['collectives_per_step'] = dist_utils.summarize_collectives(dist_utils.COLLECTIVE_LOG)
    dist_utils.COLLECTIVE_LOG = None
    # ... and one with HIP events around every collective (round 6: the first scaling curve has to say
    # where its missing fraction went): `exposed_syncbn_ms` = time the compute stream was held by the
    # SyncBN sums; `grad_bucket_ms` = busy time of the gradient buckets on their side stream (hidden
    # under the backward pass unless `finish_wait_ms` -- how long Adam waited for the last ones -- says
    # otherwise); `finish_buckets` = buckets the drip pacing had not sent by the end of the backward pass
    dist_utils.COLLECTIVE_EVENTS = []
    t1 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    inst_ms = 1e3 * (time.perf_counter() - t1)
    ev = dist_utils.summarize_collective_events(dist_utils.COLLECTIVE_EVENTS)
    dist_utils.COLLECTIVE_EVENTS = None
    sync = getattr(gan, '_sync', None)
    out['collectives_per_step']['device'] = {
        'exposed_syncbn_ms': sum(v['ms'] for k, v in ev.items() if k.startswith('syncbn')),
        'syncbn_us_each': 1e3 * sum(v['ms'] for k, v in ev.items() if k.startswith('syncbn')) /
                          max(1, sum(v['count'] for k, v in ev.items() if k.startswith('syncbn'))),
        'grad_bucket_ms': ev.get('grad_bucket', {}).get('ms', 0.0),
        'finish_wait_ms': ev.get('finish_wait', {}).get('ms', 0.0),
        'finish_buckets': getattr(sync, 'last_finish_buckets', None),
        'instrumented_step_ms': inst_ms,
        'by_kind': ev}
    # every rank's own time for the timed steps (the line's `ms_per_step` is their maximum)
    import torch.distributed as dist
    mine = torch.tensor([1e3 * own_dt / args.steps], dtype=torch.float64, device=dev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    out['ms_per_step_by_rank'] = [float(t.item()) for t in allr]
  if world == 1 and not getattr(args, 'no_shipped', False):
    out['shipped_schedule'] = shipped_schedule(gan, n, h, dev, rank)
  if world == 1 and args.batch <= 0 and h == 512 and not getattr(args, 'no_batch_max', False):
    # "N = B (largest that fits; report B)".  The headline stays at the named
    # per-GPU batch 8 (= cfg4's share of global batch 64); the large-batch rate rides along.
    del batch
    out['batch_max'] = batch_max(gan, h, dev, rank)
  if world == 1 and args.dtype == 'bf16' and not getattr(args, 'no_fp32', False):
    # (after batch_max: the fp32 path leaves its operand copies and larger wgrad slabs allocated)
    out['fp32_step'] = fp32_step(gan, h, dev, rank)
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    out['cpu_baseline'] = cpu_baseline(args, gan, summ['flops'] / 1e12 / n)
  return out


def shipped_schedule(gan, n, h, dev, rank, steps=3):
  """SURVEY 8d's secondary figure: the SHIPPED schedule, `d_step_per_g_step = 2`
  (configs/highres/highres.gin:13).  One cluster step (gan_manager.py:376-385) splits the cluster
  batch into two chunks: `train_d` on the first, `train_g_d` on the second; panoramas/s = samples
  consumed by the cluster step (2 x per-GPU batch) / its wall time."""
  import itertools
  keep = gan.d_step_per_g_step, gan.train_ds
  a, b = synth_batch(n, h, 1234 + rank, dev), synth_batch(n, h, 9876 + rank, dev)
  cluster = {k: torch.cat([b[k], a[k]]) for k in a}   # train_d sees chunk 0, train_g_d chunk 1
  del a, b
  try:
    gan.d_step_per_g_step = 2
    gan.train_ds = itertools.repeat(cluster)
    gan.train_cluster(1)
    gan.global_step += gan.num_batched_steps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
      gan.train_cluster(1)
      gan.global_step += gan.num_batched_steps
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
  finally:
    gan.d_step_per_g_step, gan.train_ds = keep
  return {'d_step_per_g_step': 2, 'value': 2 * n * steps / dt, 'unit': 'panoramas/sec',
          'ms_per_cluster_step': 1e3 * dt / steps, 'cluster_batch': 2 * n, 'steps': steps,
          'warmup': 1, 'what': 'train_d (chunk 0) + train_g_d (chunk 1) per cluster step'}


def fp32_step(gan, h, dev, rank, nb=2, steps=2):
  """The SAME step in the reference's own arithmetic: the reference trains in fp32 throughout
  (trainers/gan_manager.py:175-183, no mixed-precision policy anywhere); BASELINE.json's cfg3 names
  bf16, which the headline measures.  Here the models' compute dtype is switched to fp32 -- the path
  that carries the 1e-3 parity (fp32-input MFMA, 157.3 TFLOP/s dense peak) -- for one warm-up and
  `steps` timed train_g_d steps at per-GPU batch `nb`, then one instrumented step for the conv
  family's fraction of that peak.  The bf16 operand copies are refreshed lazily afterwards."""
  models = (gan.generator, gan.discriminator, gan.ema_generator)
  keep = [m.dtype for m in models]
  import gc
  gc.collect()
  torch.cuda.empty_cache()
  try:
    for m in models:
      m.dtype = torch.float32
    batch = synth_batch(nb, h, 777 + rank, dev)
    def step():
      gan.train_g_d(batch)
      gan.global_step += gan.num_batched_steps
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
      step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = nn.ConvProfiler()
    nn.set_conv_profiler(prof)
    step()
    torch.cuda.synchronize()
    nn.set_conv_profiler(None)
    summ = prof.summary()
    ach = summ['flops'] / (summ['ms'] * 1e-3) / 1e12 if summ['ms'] > 0 else 0.0
    return {'dtype': 'f32', 'per_gpu_batch': nb, 'value': nb * steps / dt, 'unit': 'panoramas/sec',
            'ms_per_step': 1e3 * dt / steps, 'steps': steps, 'warmup': 1,
            'roofline': {'bound': 'mfma', 'achieved': ach, 'peak': F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': ach / F32_PEAK_TFLOPS, 'conv_ms_per_step': summ['ms'],
                         'conv_tflop_per_step': summ['flops'] / 1e12,
                         'frac_in_step': summ['flops'] / (1e-3 * (1e3 * dt / steps)) / 1e12 / F32_PEAK_TFLOPS},
            'what': 'the reference trains in fp32 (gan_manager.py:175-183): same model, same step, '
                    'compute dtype fp32 (v_mfma_f32_32x32x2_f32, the 1e-3 parity path)'}
  except (torch.OutOfMemoryError, RuntimeError) as e:
    nn.set_conv_profiler(None)
    return {'dtype': 'f32', 'value': None, 'error': repr(e)[:200]}
  finally:
    for m, d in zip(models, keep):
      m.dtype = d
    gc.collect()
    torch.cuda.empty_cache()


def batch_max(gan, h, dev, rank, candidates=(32, 24, 16), steps=3):
  """Throughput of the same step at the largest per-GPU batch that fits (the batch-independent
  part -- optimiser, operand staging, launch ramps -- is amortised): two warm-up + `steps` timed
  steps, then (round 6) one instrumented step for the conv family's own roofline at that batch.
  Falls back to the next candidate when the allocation fails.  Batch 32 fits since round 6 was
  measured (265 GiB of the 288 GB): SURVEY 8d's "cfg3 N = B, largest that fits; report B"."""
  import gc
  degraded = False   # a candidate failed mid-step: later figures come from a half-stepped model
  for nb in candidates:
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    try:
      big = synth_batch(nb, h, 4321 + rank, dev)
      for _ in range(2):   # (two warm-up steps: the allocator regrows its pools after empty_cache)
        gan.train_g_d(big)
        gan.global_step += gan.num_batched_steps
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(steps):
        gan.train_g_d(big)
        gan.global_step += gan.num_batched_steps
      torch.cuda.synchronize()
      dt = time.perf_counter() - t0
      res = {'per_gpu_batch': nb, 'value': nb * steps / dt, 'unit': 'panoramas/sec',
             'ms_per_step': 1e3 * dt / steps, 'steps': steps, 'warmup': 2,
             'hbm_gib_peak': torch.cuda.max_memory_allocated(dev) / 2**30,
             'degraded': degraded, 'roofline': None}
      try:   # the conv family alone at this batch (serial instrumented step, as the main line's `frac`)
        for _ in range(2):
          prof = nn.ConvProfiler()
          nn.set_conv_profiler(prof)
          gan.train_g_d(big)
          gan.global_step += gan.num_batched_steps
          torch.cuda.synchronize()
          nn.set_conv_profiler(None)
        summ = prof.summary()
        ach = summ['flops'] / (summ['ms'] * 1e-3) / 1e12
        res['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                           'frac': ach / BF16_PEAK_TFLOPS, 'conv_ms_per_step': summ['ms'],
                           'conv_tflop_per_step': summ['flops'] / 1e12,
                           'frac_in_step': summ['flops'] / (dt / steps) / 1e12 / BF16_PEAK_TFLOPS}
        res['hbm_gib_peak'] = torch.cuda.max_memory_allocated(dev) / 2**30
      except (torch.OutOfMemoryError, RuntimeError) as e:   # (the serial path's extra workspaces did not fit)
        nn.set_conv_profiler(None)
        res['roofline_error'] = repr(e)[:200]
        torch.cuda.synchronize()
      if nb > 24 and 24 in candidates:
        # ... and batch 24 beside it (rounds 3-5's figure): at 258 of the 268 GiB the allocator works
        # harder, and the rate saturates near batch 24 -- the line says so instead of hiding it
        try:
          big = None
          gc.collect()
          torch.cuda.empty_cache()
          b24 = synth_batch(24, h, 4321 + rank, dev)
          for _ in range(2):
            gan.train_g_d(b24)
            gan.global_step += gan.num_batched_steps
          torch.cuda.synchronize()
          t0 = time.perf_counter()
          for _ in range(steps):
            gan.train_g_d(b24)
            gan.global_step += gan.num_batched_steps
          torch.cuda.synchronize()
          dt24 = time.perf_counter() - t0
          res['at_batch_24'] = {'value': 24 * steps / dt24, 'unit': 'panoramas/sec',
                                'ms_per_step': 1e3 * dt24 / steps, 'steps': steps, 'warmup': 2}
        except (torch.OutOfMemoryError, RuntimeError) as e:
          res['at_batch_24'] = {'error': repr(e)[:200]}
          torch.cuda.synchronize()
      return res
    except (torch.OutOfMemoryError, RuntimeError) as e:   # does not fit: next candidate
      last = repr(e)[:200]
      big = None
      degraded = True
      torch.cuda.synchronize()
  return {'per_gpu_batch': None, 'value': None, 'error': last, 'degraded': True}


def cpu_baseline(args, gan, tflop_per_sample, budget_s=46.0):
  """The reference's CPU path beside the GPU number: TensorFlow cannot run here, so
  this is the PyTorch-CPU restatement of the SAME step -- oracle/nets_torch.train_g_d: generator
  and discriminator forward, both backward passes, losses, per-tensor clip -- on the bench's own
  weights (copied off the device), fp32.

  `value` is MEASURED AT THE METRIC'S SHAPE: one batch-1 `train_g_d` at the benchmarked resolution
  (512x1024; SURVEY 8d: "time cfg3 for 1-2 steps").  No warm-up run: a warm-up changes nothing here
  (30.6 s against 30.7 s) and a step costs half the budget.  A CPU step is
  dominated by work that does NOT scale with pixels (W / sigma and its gradient over 1.1 B
  parameters, several weight-sized fp32 passes per layer), so batch 1 is the cheapest sample that
  still is the workload; the rate at batch 8 would be higher by the amortised weight-bound part,
  and the secondary run at cfg1's resolution (128x256, `cfg1`) keeps the earlier rounds' figure
  comparable.  A port, not TF; baseline only."""
  import torch as T
  from oracle import nets_torch as O
  try:
    cores = len(os.sched_getaffinity(0))
  except AttributeError:
    cores = os.cpu_count() or 1
  # 16 threads: measured FASTEST on the GPU boxes' 2 x 64-core hosts (tools/oracle_f64_time.py:
  # the same oracle pass takes 9.9 s on 16 threads, 26 s on 64, 75 s on 128 -- weight-sized passes
  # and small-M products that more threads only oversubscribe); the baseline gets the setting
  # that is best for it, `cores` reports it
  threads = max(1, min(cores, 16))
  T.set_num_threads(threads)
  t_start = time.perf_counter()
  cpu = lambda m: {k: v.detach().cpu() for k, v in m.store.views.items()}
  gp, dp = cpu(gan.generator), cpu(gan.discriminator)
  cfg = dict(gen=dict(gen_dims=gan.generator.hidden_dims, resnet_version=gan.generator.resnet_version,
                      context_layer='convs', z_dim=gan.generator.z_dim),
             dis=dict(n_dis=len(gan.discriminator.discriminators),
                      n_layers=len(gan.discriminator.discriminators[0].groups) + 1, kernel_size=4),
             lambda_gan=gan.lambda_gan, lambda_kld=gan.lambda_kld, lambda_wc=gan.lambda_wc,
             lambda_depth=gan.lambda_depth, mask_blurred=gan.mask_blurred,
             g_train=lambda k: not k.endswith(('/u', '/moving_mean', '/moving_variance')),
             d_train=lambda k: not k.endswith('/u'))
  g = T.Generator().manual_seed(1234)
  def timed(n, h):
    b = synth_batch_cpu(n, h, g)
    t0 = time.perf_counter()
    O.train_g_d(gp, dp, b, cfg)
    return time.perf_counter() - t0
  h_hi = args.image_size
  t_hi = timed(1, h_hi)
  out = {'value': 1.0 / t_hi, 'unit': 'panoramas/sec', 'cores': threads, 'kind': 'port',
         'resolution': f'{h_hi}x{2 * h_hi}', 'batch': 1, 'timed_runs_s': [t_hi], 'cfg1': None}
  note = ''
  # secondary: the cfg1 resolution (BASELINE.json configs[0]) at batch 1, if the budget allows (a
  # step there costs ~10 s -- the weight-bound part -- against ~34 s at 512x1024)
  h_lo, n_lo = 128, 1
  if h_hi != h_lo and budget_s - (time.perf_counter() - t_start) > 10.5:
    t_lo = timed(n_lo, h_lo)
    out['cfg1'] = {'value': n_lo / t_lo, 'unit': 'panoramas/sec', 'resolution': f'{h_lo}x{2 * h_lo}',
                   'batch': n_lo, 'timed_runs_s': [t_lo]}
    note = (f'  Secondary (cfg1 shape {h_lo}x{2 * h_lo} batch {n_lo}): {t_lo:.1f} s = '
            f'{n_lo / t_lo:.4f} panoramas/s at that size.')
  out['leg_s'] = time.perf_counter() - t_start
  out['sample'] = (f'ONE measured oracle.nets_torch.train_g_d step (PyTorch-CPU fp32 restatement of '
                   f'the full G+D step: forward, both backward passes, losses, clip; the bench '
                   f'weights) at the benchmarked resolution {h_hi}x{2 * h_hi}, batch 1: {t_hi:.1f} s '
                   f'= {1.0 / t_hi:.4f} panoramas/s on {threads} host threads.' + note + '  Not TF.')
  return out


def synth_batch_cpu(n, h, g):
  import torch as T
  w = 2 * h
  r = lambda *s: T.rand(s, generator=g)
  image, depth, poison = r(n, h, w, 3), r(n, h, w, 1), r(n, h, w, 1)
  depth = T.where(poison < 0.02, T.zeros_like(depth), depth)
  depth = T.where(poison > 0.99, T.ones_like(depth), depth)
  pm = (r(n, h, w, 1) < 0.5).float()
  pm[:, h // 3:h // 3 + max(1, h // 8)] = 0
  bm = T.zeros((n, h, w, 1))
  bm[:, :h // 8] = 1
  bm[:, -(h // 8):] = 1
  return dict(image=image, depth=depth, proj_mask=pm, proj_image=image * pm, proj_depth=depth * pm,
              blurred_mask=bm)
