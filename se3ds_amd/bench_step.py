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
  n = args.batch if args.batch > 0 else 4
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
      'roofline': {'bound': 'mfma', 'kernel': 'igemm_kernel/wgrad_kernel (all conv fwd+dgrad+wgrad '
                   'launches of one step)', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                   'frac': achieved / peak, 'traffic': None, 'launches': summ['launches'],
                   'avg_launch_ms': summ['ms'] / max(summ['launches'], 1),
                   'conv_ms_per_step': summ['ms'], 'conv_tflop_per_step': summ['flops'] / 1e12,
                   'by_kind': summ['by_kind']},
      'losses': {k: float(v) for k, v in m.items() if k in (
          'gen/gen_gan_loss', 'dis/disc_loss', 'gen/depth_loss', 'gen/wc_loss')},
      'hbm_gib_peak': torch.cuda.max_memory_allocated(dev) / 2**30,
  }
  if rank == 0 and not args.no_cpu_baseline:
    out['cpu_baseline'] = cpu_baseline(args)
  return out


def cpu_baseline(args):
  """The oracle (PyTorch-CPU fp32 restatement of the same step -- TF cannot run here) on a
  bounded sample: ONE train_g_d sample at 128x256 with the full ResNet-101 / gen_dims-128
  network, scaled to 512x1024 by the conv-FLOP ratio (16x)."""
  import torch as T
  from oracle import nets_torch as O
  T.set_num_threads(os.cpu_count() or 1)
  h = 128
  gin_lite.clear_config()
  G = image_models.ResNetGenerator(image_size=h, gen_dims=128, resnet_version='101', device='cpu',
                                   seed=None)
  D = image_models.SNMultiScaleDiscriminator(dis_dims=128, n_layers=6, n_dis=2, device='cpu',
                                             seed=None)
  gp = {k: v.detach() for k, v in G.store.views.items()}
  dp = {k: v.detach() for k, v in D.store.views.items()}
  batch = {k: v.cpu() for k, v in synth_batch(1, h, 1, 'cpu').items()}
  cfg = dict(gen=dict(gen_dims=128, resnet_version='101', context_layer='convs', z_dim=128),
             dis=dict(n_dis=2, n_layers=6, kernel_size=4), lambda_gan=1.0, lambda_kld=10.0,
             lambda_wc=10.0, lambda_depth=100.0, mask_blurred=True,
             g_train=lambda k: not k.endswith(('/u', '/moving_mean', '/moving_variance')),
             d_train=lambda k: not k.endswith('/u'))
  t0 = time.perf_counter()
  O.train_g_d(gp, dp, batch, cfg)
  dt = time.perf_counter() - t0
  scale = (512 * 1024) / (h * 2 * h)
  return {'value': 1.0 / (dt * scale), 'unit': 'panoramas/sec', 'cores': os.cpu_count(),
          'kind': 'port',
          'sample': f'1 train_g_d sample (fwd+bwd, no optimizer) at {h}x{2 * h}, full ResNet-101 '
                    f'gen_dims=128 network, PyTorch-CPU fp32 oracle (not TF): {dt:.1f} s, scaled '
                    f'x{scale:.0f} to 512x1024 by pixel count'}
