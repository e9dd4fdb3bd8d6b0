"""Which normalisation layers of the benchmarked step still take a statistics pass of their OWN
(forward: not from the producing convolution's epilogue), and which backward path each norm takes:
one `train_g_d` of the bench configuration under nn._NORM_DEBUG, printed by tensor size.

  python tools/norm_audit.py [--batch 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3ds_amd import bench_step  # noqa: E402
from se3ds_amd.hipops import nn  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=8)
  ap.add_argument('--image-size', type=int, default=512)
  ap.add_argument('--dtype', default='bf16')
  args = ap.parse_args()
  dev = torch.device('cuda:0')
  gan = bench_step.build_gan(args, dev, 1)
  batch = bench_step.synth_batch(args.batch, args.image_size, 1234, dev)
  for it in range(2):
    nn._NORM_DEBUG = {} if it == 1 else None
    gan.train_g_d(batch)
    gan.global_step += gan.num_batched_steps
  torch.cuda.synchronize()
  dbg, nn._NORM_DEBUG = nn._NORM_DEBUG, None
  own, other = [], {}
  for k, v in dbg.items():
    if k[0] in ('cg', 'fused-rows', 'fused-bwd'):
      other.setdefault(k[0], []).append((k[1], k[2], v))
    else:
      shape = k[1]
      mb = 2 * int(torch.tensor(shape).prod()) / 1e6
      own.append((mb, k[0], shape, k[2], v))
  print('forward statistics passes of their own (not from a conv epilogue):')
  tot = 0.0
  for mb, name, shape, kind, v in sorted(own, reverse=True):
    tot += mb * v
    print(f'  {mb:8.1f} MB x{v}  {kind:8s} {str(shape):28s} {name}')
  print(f'  total {tot / 1e3:.2f} GB read per step by {sum(o[4] for o in own)} passes')
  for tag, rows in other.items():
    print(f'backward path {tag}: {sum(r[2] for r in rows)} norms')


if __name__ == '__main__':
  main()
