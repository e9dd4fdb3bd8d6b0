"""Wall time of the oracle's binary64 generator forward + backward at the yardstick test's size
(64x128, batch 1, real widths) on the host cores -- the cost that dominates the GPU suite.
SE3DS_ORACLE_F64_DIV=kernel|output selects where the spectral division runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import nets_torch as O
from se3ds_amd.models import image_models
from se3ds_amd.bench_step import synth_batch
G = image_models.ResNetGenerator(image_size=64, gen_dims=128, resnet_version='101', device='cpu', seed=-3)
names = set(G.store.trainable_names)
snap = {k: v.clone() for k, v in G.store.views.items()}
batch = {k: v.cpu() for k, v in synth_batch(1, 64, 55, torch.device('cpu')).items()}
for dt in (torch.float32, torch.float64):
  torch.set_default_dtype(dt)
  p = {k: v.to(dt if v.is_floating_point() else v.dtype).clone().requires_grad_(k in names) for k, v in snap.items()}
  b = {k: v.to(dt) for k, v in batch.items()}
  t0 = time.time()
  outs, _ = O.generator_forward(p, b, True, gen_dims=128, resnet_version='101', z_dim=128, bn_training=False)
  t1 = time.time()
  (outs[6].sum() + outs[3].sum()).backward()
  print(f'{dt} threads {torch.get_num_threads()} div={os.environ.get("SE3DS_ORACLE_F64_DIV", "output")}: '
        f'fwd {t1 - t0:.1f} s, bwd {time.time() - t1:.1f} s', flush=True)
  del p, outs
