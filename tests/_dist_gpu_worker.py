"""Worker of tests/test_dist_gpu.py: two ranks share cuda:0 and talk over gloo (RCCL refuses two
ranks on one device); the trainer path is the multi-replica one: SyncBN statistics all-reduce,
per-module clip + side-stream gradient all-reduce on its own process group, Adam on the sum."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3ds_amd import gin_lite  # noqa: E402
from se3ds_amd.models import image_models  # noqa: E402
from se3ds_amd.trainers import dist_utils, gan_manager, se3ds_trainer  # noqa: E402
from tests.test_nets_gpu import synth_batch  # noqa: E402


def main():
  import signal
  signal.alarm(200)   # never outlive the test: a failed peer would leave this rank in a collective
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
  batch = {k: v.to(dev) for k, v in dist_utils.shard_batch(full, rank, world).items()}
  gan.train_g_d(batch)
  gan.global_step += 1
  gan.train_d(batch)
  gan.train_g_d(batch)
  torch.cuda.synchronize()
  assert gan._sync is not None and gan._sync.group is not gan.strategy.group
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
    print('DIST_GPU_OK', [float(x) for x in got[0][:5]])
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
