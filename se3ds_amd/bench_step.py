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
  barrier(world)
  dt = max_over_ranks(time.perf_counter() - t0, world, dev)
  ms = 1e3 * dt / args.steps

  # ---- roofline of the dominant kernel family (implicit-GEMM convolutions): one extra,
  # instrumented step with HIP events around every conv launch on the launch stream
  prof = nn.ConvProfiler()
  nn.set_conv_profiler(prof)
  step()
  torch.cuda.synchronize()
  nn.set_conv_profiler(None)
  summ = prof.summary()
  if os.environ.get('SE3DS_BENCH_SHAPES'):
    rows = sorted(prof.by_shape().items(), key=lambda kv: -kv[1][0])
    for (kind, tag), (ms_, fl, cnt) in rows[:40]:
      print(f'SHAPE {ms_:8.2f} ms {fl / max(ms_, 1e-9) / 1e9:7.0f} TF/s x{cnt:3d} {kind:6s} {tag}')
  peak = BF16_PEAK_TFLOPS if args.dtype == 'bf16' else F32_PEAK_TFLOPS
  achieved = summ['flops'] / (summ['ms'] * 1e-3) / 1e12 if summ['ms'] > 0 else 0.0
  m = gan._save_metrics_to_dict()
  # HBM traffic of the dominant kernel (3x3 1024->1024 @32x64 forward, 35 % of the conv time with
  # its data-gradient twin) from the tracked PMC summary; the family is MFMA-bound, the figure
  # shows that no kernel re-reads its operands from HBM (algorithmic: 52 MB in, 33.5 MB out)
  traffic = traffic_detail = None
  if h == 512 and n == 8 and args.dtype == 'bf16':
    import json
    path = os.path.join(ROOT, 'profiles', 'r02_conv_pmc_1024_1024_3_1_32_64_1_8.json')
    if os.path.exists(path):
      k = json.load(open(path)).get('igemm_halo_kernel<0, 256, 2>')
      if k and 'fetch_mb' in k:
        traffic = (k['fetch_mb'] + k['write_mb']) * 1e6
        traffic_detail = {'source': 'profiles/r02_conv_pmc_1024_1024_3_1_32_64_1_8.json',
                          'kernel': 'igemm_halo_kernel<0, 256, 2> (3x3 1024->1024 @32x64, batch 8)',
                          'fetch_mb_raw': k['fetch_mb'],
                          'fetch_mb_x2_wide_read_correction': 2 * k['fetch_mb'],
                          'write_mb': k['write_mb'], 'algorithmic_mb': 52.4 + 33.6}
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
                   'frac': achieved / peak, 'traffic': traffic, 'traffic_detail': traffic_detail,
                   'launches': summ['launches'],
                   'avg_launch_ms': summ['ms'] / max(summ['launches'], 1),
                   'conv_ms_per_step': summ['ms'], 'conv_tflop_per_step': summ['flops'] / 1e12,
                   'by_kind': summ['by_kind']},
      'losses': {k: float(v) for k, v in m.items() if k in (
          'gen/gen_gan_loss', 'dis/disc_loss', 'gen/depth_loss', 'gen/wc_loss')},
      'hbm_gib_peak': torch.cuda.max_memory_allocated(dev) / 2**30,
  }
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    out['cpu_baseline'] = cpu_baseline(args, gan, summ['flops'] / 1e12 / n)
  return out


def cpu_baseline(args, gan, tflop_per_sample):
  """The reference's CPU path beside the GPU number: TensorFlow cannot run here, so
  this is the PyTorch-CPU restatement of the SAME step -- oracle/nets_torch.train_g_d: generator
  and discriminator forward, both backward passes, losses, per-tensor clip -- on the bench's own
  weights (copied off the device), all host cores, fp32, at cfg1's resolution (128x256, batch 1):
  a bounded sample (~10-30 s).  The convolutional work per panorama scales with the pixel count,
  so the 512x1024 rate is the measured 128x256 rate / 16; a conv-only extrapolation from the
  dominant 512x1024 layer shape is kept as a cross-check.  A port, not TF; baseline only."""
  import torch as T
  from oracle import nets_torch as O
  try:
    cores = len(os.sched_getaffinity(0))
  except AttributeError:
    cores = os.cpu_count() or 1
  threads = max(1, min(cores, 64))
  T.set_num_threads(threads)
  cpu = lambda m: {k: v.detach().cpu() for k, v in m.store.views.items()}
  gp, dp = cpu(gan.generator), cpu(gan.discriminator)
  h_lo = 128
  g = T.Generator().manual_seed(1234)
  batch = {k: v.cpu() for k, v in synth_batch_cpu(1, h_lo, g).items()}
  cfg = dict(gen=dict(gen_dims=gan.generator.hidden_dims, resnet_version=gan.generator.resnet_version,
                      context_layer='convs', z_dim=gan.generator.z_dim),
             dis=dict(n_dis=len(gan.discriminator.discriminators),
                      n_layers=len(gan.discriminator.discriminators[0].groups) + 1, kernel_size=4),
             lambda_gan=gan.lambda_gan, lambda_kld=gan.lambda_kld, lambda_wc=gan.lambda_wc,
             lambda_depth=gan.lambda_depth, mask_blurred=gan.mask_blurred,
             g_train=lambda k: not k.endswith(('/u', '/moving_mean', '/moving_variance')),
             d_train=lambda k: not k.endswith('/u'))
  t0 = time.perf_counter()
  O.train_g_d(gp, dp, batch, cfg)
  dt = time.perf_counter() - t0
  scale = (args.image_size / h_lo) ** 2
  # cross-check: conv fwd+bwd FLOP rate of the dominant layer shape -> conv-only panoramas/s
  x = T.randn((1, 32, 64, 1024), generator=g, requires_grad=True)
  wgt = T.randn((3, 3, 1024, 1024), generator=g, requires_grad=True)
  t1 = time.perf_counter()
  reps = 0
  while reps < 1 or (time.perf_counter() - t1 < 3.0 and reps < 50):
    y = O.tf_conv2d(O.pad_layer(x, 1), wgt, 1, 'VALID')
    y.backward(T.ones_like(y))
    x.grad = wgt.grad = None
    reps += 1
  rate = reps * 3 * 2.0 * 32 * 64 * 1024 * 1024 * 9 / (time.perf_counter() - t1)
  return {'value': 1.0 / (dt * scale), 'unit': 'panoramas/sec', 'cores': threads, 'kind': 'port',
          'sample': f'oracle.nets_torch.train_g_d (PyTorch-CPU fp32 restatement of the full G+D '
                    f'step: forward, both backward passes, losses, clip; same weights) on ONE '
                    f'{h_lo}x{2 * h_lo} panorama: {dt:.1f} s = {1.0 / dt:.4f} panoramas/s at that size; '
                    f'/{scale:.0f} (pixel ratio) for {args.image_size}x{2 * args.image_size}.  Cross-check, '
                    f'conv-only: 3x3 1024->1024@32x64 fwd+bwd at {rate / 1e12:.3f} TFLOP/s -> '
                    f'{rate / (tflop_per_sample * 1e12):.4f} panoramas/s.  Not TF.',
          'measured_lowres_panoramas_per_sec': 1.0 / dt}


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
