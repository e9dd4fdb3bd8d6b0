"""Worker of tests/test_dist_gpu.py: two ranks share cuda:0 and talk over gloo (RCCL refuses two
ranks on one device).  The trainer path is the multi-replica one -- SyncBN statistics all-reduce,
per-module clip + side-stream gradient all-reduce, Adam on the sum -- and the PRODUCT's summed,
per-replica-clipped gradient is compared with the R-replica oracle (statistics pooled
over the replicas as SyncBatchNormalization does, per-replica loss / backward / per-tensor clip,
then the sum; se3ds_trainer.py:230-257)."""
import faulthandler
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nets_torch as O  # noqa: E402
from se3ds_amd import gin_lite  # noqa: E402
from se3ds_amd.models import image_models  # noqa: E402
from se3ds_amd.trainers import dist_utils, gan_manager, se3ds_trainer  # noqa: E402
from tests.test_nets_gpu import synth_batch  # noqa: E402


def _oracle_sum(gp, dp, shard, world, double):
  cfg = dict(gen=dict(gen_dims=8, resnet_version='50', context_layer='convs', z_dim=4,
                      stats_hook=O.pooled_stats_hook(world)),
             dis=dict(n_dis=2, n_layers=3, kernel_size=4), lambda_gan=1.0, lambda_kld=10.0,
             lambda_wc=10.0, lambda_depth=100.0, mask_blurred=True,
             g_train=lambda k: not k.endswith(('/u', '/moving_mean', '/moving_variance')),
             d_train=lambda k: not k.endswith('/u'))
  if double:
    torch.set_default_dtype(torch.float64)
    conv = lambda d: {k: (v.double() if v.is_floating_point() else v) for k, v in d.items()}
    gp, dp, shard = conv(gp), conv(dp), conv(shard)
  try:
    ref = O.train_g_d(gp, dp, shard, cfg, replicas=world)
  finally:
    torch.set_default_dtype(torch.float32)
  out = {}
  for key in ('g_grads', 'd_grads'):
    names = list(ref[key])
    flat = torch.cat([ref[key][k].reshape(-1) for k in names])
    assert max(float(ref[key][k].norm()) for k in names) <= 5.0 + 1e-4   # clipped per replica
    dist.all_reduce(flat)   # SUM of the per-replica clipped gradients
    out[key] = (names, [tuple(ref[key][k].shape) for k in names], flat)
  out['mm'] = ref['g_updates']['encoder/bn1/moving_mean']
  return out


def main():
  # a hang must be visible: dump every thread's stack, then die with a non-zero status
  faulthandler.enable()
  faulthandler.dump_traceback_later(170, exit=True)
  dist.init_process_group('gloo')
  rank, world = dist.get_rank(), dist.get_world_size()
  dev = 'cuda:0'
  torch.cuda.set_device(0)
  gin_lite.clear_config()
  gin_lite.parse_config('''
image_models.ResNetGenerator.gen_dims = 8
image_models.ResNetGenerator.z_dim = 4
image_models.ResNetGenerator.resnet_version = "50"
image_models.SNMultiScaleDiscriminator.dis_dims = 4
image_models.SNMultiScaleDiscriminator.n_dis = 2
image_models.SNMultiScaleDiscriminator.n_layers = 3
''')
  gan = se3ds_trainer.GAN(
      strategy=gan_manager.DataParallelStrategy(dev), model_dir='', lambda_gan=1.0, lambda_kld=10.0,
      lambda_wc=10.0, lambda_depth=100.0, mask_blurred=True, predict_depth=True, image_size=64,
      beta1=0.5, g_lr=1e-4, d_lr=4e-4, d_step_per_g_step=1, num_batched_steps=1,
      generator_fn=image_models.ResNetGenerator,
      discriminator_fn=image_models.SNMultiScaleDiscriminator, seed=0, compute_dtype=torch.float32)
  gan._create_obj()
  full = synth_batch(2 * world, 64, seed=91)
  shard = dist_utils.shard_batch(full, rank, world)
  batch = {k: v.to(dev) for k, v in shard.items()}

  # ---- R-replica oracle (fp32 and the fp64 yardstick) on the initial weights
  gp = {k: v.detach().cpu().clone() for k, v in gan.generator.store.views.items()}
  dp = {k: v.detach().cpu().clone() for k, v in gan.discriminator.store.views.items()}
  ref32 = _oracle_sum(gp, dp, shard, world, False)
  ref64 = _oracle_sum(gp, dp, shard, world, True)

  # ---- product step; capture the summed gradient arenas as Adam consumes them
  from tests.test_nets_gpu import capture_clipped_grads
  cap = {tag: capture_clipped_grads(opt)[0]
         for opt, tag in ((gan.g_optimizer, 'g_grads'), (gan.d_optimizer, 'd_grads'))}
  gan.train_g_d(batch)
  torch.cuda.synchronize()
  # the two decoders (+ heads) run in lockstep and their k-th SyncBatchNormalization
  # sums share one all-reduce, forward and backward: a step issues 2 x (all batch norms - those of
  # one decoder branch) collectives instead of 2 x all
  gnames = gan.generator.store.trainable_names
  n_bn = sum(1 for k in gnames if k.endswith('/gamma'))
  n_branch = sum(1 for k in gnames if k.endswith('/gamma') and k.startswith(('decoder/', 'rgb_conv/')))
  if rank == 0:
    print(f'SyncBN all-reduces per step: {gan.last_collectives} (unpaired: {2 * n_bn}; '
          f'{n_bn} batch norms, {n_branch} per decoder branch)')
  assert n_branch > 0.25 * n_bn
  assert gan.last_collectives == 2 * (n_bn - n_branch), (gan.last_collectives, n_bn, n_branch)
  for key, model in (('g_grads', gan.generator), ('d_grads', gan.discriminator)):
    names, shapes, f32 = ref32[key]
    _, _, f64 = ref64[key]
    st = model.store
    assert set(names) == set(st.trainable_names)
    hip = torch.cat([cap[key][st._off_tr[k][0]:st._off_tr[k][0] + st._off_tr[k][1]] for k in names])
    e_hip = float((hip.double() - f64).norm() / f64.norm())
    e_o32 = float((f32.double() - f64).norm() / f64.norm())
    if rank == 0:
      print(f'R={world} {key}: ||sum grad - f64|| / ||f64||: hip {e_hip:.3e}, fp32 oracle {e_o32:.3e}')
    assert e_hip <= 5.0 * e_o32 + 1e-3, (key, e_hip, e_o32)
    o = 0
    gn = float(f64.norm())
    for k, shp in zip(names, shapes):
      cnt = int(np.prod(shp)) if len(shp) else 1
      a, b32, b64 = hip[o:o + cnt].double(), f32[o:o + cnt].double(), f64[o:o + cnt]
      o += cnt
      if float(b64.norm()) < 1e-3 * gn:
        continue   # tensors that carry none of the gradient: noise only
      ea, eo = float((a - b64).norm() / b64.norm()), float((b32 - b64).norm() / b64.norm())
      assert ea <= 5.0 * eo + 2e-3, (key, k, ea, eo)
  # pooled batch statistics: the moving mean equals the R-replica oracle's
  mm = gan.generator.store['encoder/bn1/moving_mean'].cpu().double()
  assert float((mm - ref64['mm']).abs().max()) <= 1e-5 * float(ref64['mm'].abs().max()) + 1e-7

  gan.global_step += 1
  gan.train_d(batch)
  gan.train_g_d(batch)
  torch.cuda.synchronize()
  own = os.environ.get('SE3DS_GRAD_SYNC_OWN_COMM') == '1'
  assert gan._sync is not None and (gan._sync.group is not gan.strategy.group) == own
  vecs = [gan.generator.store.theta, gan.discriminator.store.theta, gan.g_optimizer.v,
          gan.d_optimizer.m, gan.ema_generator.store.theta]
  sig = torch.stack([v.double().sum() for v in vecs] + [v.double().abs().sum() for v in vecs]).cpu()
  assert bool(torch.isfinite(sig).all())
  got = [torch.zeros_like(sig) for _ in range(world)]
  dist.all_gather(got, sig)
  for g in got[1:]:
    # replicas stay bit-identical: every cross-replica exchange is a deterministic sum
    np.testing.assert_array_equal(got[0].numpy(), g.numpy())
  # the replicas saw different samples: local (pre-sync) statistics must differ
  m = gan._save_metrics_to_dict()
  loss = torch.tensor([float(m['gen/depth_loss'])], dtype=torch.float64)
  ls = [torch.zeros_like(loss) for _ in range(world)]
  dist.all_gather(ls, loss)
  assert abs(float(ls[0]) - float(ls[1])) > 0, 'both ranks report the same local loss'
  if rank == 0:
    ctx_probe = gan.generator.make_ctx(training=True, record=False, group=gan.strategy.group, world=world)
    print('streams=%d' % (0 if ctx_probe.streams is None else len(ctx_probe.streams)))
    print('DIST_GPU_OK', [float(x).hex() for x in got[0]])
  faulthandler.cancel_dump_traceback_later()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
