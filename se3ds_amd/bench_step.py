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
                   'frac': achieved / peak, 'traffic': None, 'launches': summ['launches'],
                   'avg_launch_ms': summ['ms'] / max(summ['launches'], 1),
                   'conv_ms_per_step': summ['ms'], 'conv_tflop_per_step': summ['flops'] / 1e12,
                   'by_kind': summ['by_kind']},
      'losses': {k: float(v) for k, v in m.items() if k in (
          'gen/gen_gan_loss', 'dis/disc_loss', 'gen/depth_loss', 'gen/wc_loss')},
      'hbm_gib_peak': torch.cuda.max_memory_allocated(dev) / 2**30,
  }
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    out['cpu_baseline'] = cpu_baseline(args, summ['flops'] / 1e12 / n)
  return out


def cpu_baseline(args, tflop_per_sample):
  """CPU baseline on a BOUNDED sample (~20 s): the PyTorch-CPU oracle's convolution
  (oracle/nets_torch.tf_conv2d: the restatement of tf.nn.conv2d, fp32, oneDNN) forward +
  backward on the three layer shapes that carry most of the step's FLOPs, at batch 1.  The
  measured FLOP rate is converted to panoramas/s with the step's algorithmic conv FLOPs per
  sample.  A port (TensorFlow cannot run here), not the reference; conv-only, so it flatters
  the CPU (norms / optimiser are not charged)."""
  import torch as T
  from oracle import nets_torch as O
  try:
    cores = len(os.sched_getaffinity(0))
  except AttributeError:
    cores = os.cpu_count() or 1
  threads = max(1, min(cores, 64))
  T.set_num_threads(threads)
  shapes = [  # (cin, cout, k, h, w, share of G forward FLOPs: SURVEY.md section 8d)
      (1024, 1024, 3, 32, 64, 0.592), (128, 128, 3, 256, 512, 0.184), (512, 512, 3, 32, 64, 0.059)]
  g = T.Generator().manual_seed(0)
  inv_rate = 0.0
  tot_share = sum(sh[-1] for sh in shapes)
  desc = []
  for cin, cout, k, h, w, share in shapes:
    x = T.randn((1, h, w, cin), generator=g, requires_grad=True)
    wgt = T.randn((k, k, cin, cout), generator=g, requires_grad=True)
    flops = 3 * 2.0 * h * w * cin * cout * k * k   # fwd + dgrad + wgrad
    t0 = time.perf_counter()
    reps = 0
    while reps < 1 or (time.perf_counter() - t0 < 6.0 and reps < 50):
      y = O.tf_conv2d(O.pad_layer(x, 1), wgt, 1, 'VALID')
      y.backward(T.ones_like(y))
      x.grad = None
      wgt.grad = None
      reps += 1
    dt = (time.perf_counter() - t0) / reps
    rate = flops / dt
    inv_rate += (share / tot_share) / rate
    desc.append(f'{k}x{k} {cin}->{cout}@{h}x{w}: {rate / 1e12:.3f} TFLOP/s')
  rate = 1.0 / inv_rate
  return {'value': rate / (tflop_per_sample * 1e12), 'unit': 'panoramas/sec', 'cores': threads,
          'kind': 'port',
          'sample': 'PyTorch-CPU fp32 oracle conv fwd+bwd on 3 dominant layer shapes, batch 1, '
                    '~6 s each (' + '; '.join(desc) + f'), FLOP-share-weighted rate '
                    f'{rate / 1e12:.3f} TFLOP/s / {tflop_per_sample:.1f} TFLOP conv work per sample; '
                    'conv-only (norms/optimiser not charged), not TF'}
