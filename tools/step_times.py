"""Per-step wall time and host (launch) time of train_g_d at the bench configuration."""
import sys, os, time, gc, argparse, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3ds_amd import bench_step
args = argparse.Namespace(image_size=512, dtype='bf16', batch=8, gin_bindings=[])
dev = torch.device('cuda', 0)
gan = bench_step.build_gan(args, dev, 1)
batch = bench_step.synth_batch(8, 512, 1234, dev)
if os.environ.get('NOGC'):
  gc.disable()
ts, hs = [], []
for i in range(14):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  gan.train_g_d(batch); gan.global_step += 1
  t1 = time.perf_counter()
  torch.cuda.synchronize(); ts.append(time.perf_counter() - t0); hs.append(t1 - t0)
print('wall', ' '.join(f'{t*1e3:.0f}' for t in ts))
print('host', ' '.join(f'{t*1e3:.0f}' for t in hs))
print('gc counts', gc.get_count(), 'peak GiB', torch.cuda.max_memory_allocated() / 2**30)
from se3ds_amd.hipops import nn as _nn
if _nn._NORM_DEBUG is not None:
  import collections
  per = collections.Counter()
  for (name, shape, kind), cnt in _nn._NORM_DEBUG.items():
    per[(shape, kind)] += cnt
  print('norms with their own statistics pass, launches over 14 steps by shape:')
  for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:20]:
    print(' ', v, k)
  for (name, shape, kind), cnt in sorted(_nn._NORM_DEBUG.items(), key=lambda kv: -kv[1])[:40]:
    print('   ', cnt, name, shape, kind)
