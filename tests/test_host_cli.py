"""CPU: host-side plumbing -- bench.py's rank launcher and the gin surface of the reference's own
trainer test (trainers/se3ds_trainer_test.py:70-99)."""
import os
import subprocess
import sys

import pytest

from se3ds_amd import gin_lite

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_n_launches_n_ranks(monkeypatch):
  """`bench.py --gpus N` outside torchrun starts N ranks as a CHILD process (torch.distributed.run,
  127.0.0.1 rendezvous) before anything touches the GPU and returns the child's status."""
  import bench
  calls = []

  class R:
    returncode = 7

  def fake_run(cmd, env=None, **kw):
    calls.append((cmd, env))
    return R()
  monkeypatch.setattr(subprocess, 'run', fake_run)
  monkeypatch.delenv('WORLD_SIZE', raising=False)
  monkeypatch.delenv('SE3DS_BENCH_BACKEND', raising=False)
  monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '2', '--warmup', '1'])
  # fewer GPUs than ranks: fails fast with a clear message, nothing is launched
  monkeypatch.setattr(bench.torch.cuda, 'device_count', lambda: 1)
  with pytest.raises(SystemExit) as e:
    bench.main()
  assert e.value.code == 2 and not calls
  monkeypatch.setattr(bench.torch.cuda, 'device_count', lambda: 4)
  with pytest.raises(SystemExit) as e:
    bench.main()
  assert e.value.code == 7
  cmd, env = calls[0]
  assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=4' in cmd
  assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
  assert cmd[-6:] == ['--gpus', '4', '--steps', '2', '--warmup', '1']
  assert os.path.basename(cmd[cmd.index('--gpus') - 1]) == 'bench.py'
  assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_trainer_test_gin_block_parses():
  """The minimal gin block of the reference's trainer test: every selector resolves against the
  product's configurables (GANManager / GAN / image_models.*), dataset selectors are accepted."""
  from se3ds_amd.models import image_models  # noqa: F401
  from se3ds_amd.trainers import se3ds_trainer  # noqa: F401
  gin_lite.clear_config()
  gin_lite.parse_config("""
    GANManager.generator_fn = @image_models.ResNetGenerator
    GANManager.discriminator_fn = @image_models.SNMultiScaleDiscriminator
    GANManager.log_every_steps = 1
    GANManager.save_every_steps = 1
    GANManager.eval_every_steps = 1
    GANManager.shuffle_buffer_size = 2
    GANManager.train_batch_size = 2
    GANManager.test_batch_size = 2
    GANManager.d_step_per_g_step = 1
    GANManager.num_batched_steps = 1
    GANManager.eval_size = 2
    GANManager.image_size = 128
    R2RImageDataset.image_size = 128
    R2RVideoDataset.image_size = 128
    image_models.SNMultiScaleDiscriminator.n_dis = 1
    image_models.SNMultiScaleDiscriminator.dis_dims = 2
    image_models.SNMultiScaleDiscriminator.n_layers = 2
    image_models.ResNetGenerator.gen_dims = 2
    image_models.ResNetGenerator.z_dim = 2
    image_models.ResNetGenerator.image_size = 128
    image_models.ResNetGenerator.conv_mode = 'normal'
    image_models.ResNetGenerator.context_layer = 'none'
    GAN.predict_depth = True
    se3ds_trainer.GAN.dis_use_pred_depth = True
    GAN.lambda_gan = 1
    GAN.lambda_kld = 0.05
    GAN.lambda_wc = 1.0
    GAN.lambda_depth = 1.0
    """)

  class Strategy:
    num_replicas_in_sync = 1
    group = None
    device = 'cpu'
  gan = se3ds_trainer.GAN(strategy=Strategy(), model_dir='', num_epochs=-1, eval_size=2)
  assert gan.image_size == 128 and gan.d_step_per_g_step == 1 and gan.num_batched_steps == 1
  assert gan.predict_depth is True and gan.dis_use_pred_depth is True
  assert gan.lambda_gan == 1 and gan.lambda_kld == 0.05 and gan.train_batch_size == 2
  assert gan.generator_fn is image_models.ResNetGenerator
  assert gan.discriminator_fn is image_models.SNMultiScaleDiscriminator
  gin_lite.clear_config()
