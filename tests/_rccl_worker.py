"""Worker of tests/test_dist_gpu.py::test_rccl_single_rank_api_paths: ONE rank on the real `nccl`
(RCCL) backend, initialised exactly as bench.py does it, then every collective shape the
multi-GPU step issues: barrier, MAX all-reduce of a float64 scalar (timing), SUM all-reduce of
fp32 SyncBN statistics, bucketed SUM all-reduce of arena slices on a side stream through a
cloned communicator (dist_utils.GradSync).  A one-rank all-reduce is an identity, so the values
must come back unchanged; what is checked is that RCCL accepts every call."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from se3ds_amd.trainers import dist_utils  # noqa: E402


def main():
  import signal
  signal.alarm(150)
  os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
  os.environ.setdefault('MASTER_PORT', sys.argv[1])
  os.environ['RANK'], os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'] = '0', '1', '0'
  torch.cuda.set_device(0)
  dev = torch.device('cuda', 0)
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
  dist.barrier()
  t = torch.tensor([1.25], dtype=torch.float64, device=dev)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  assert float(t.item()) == 1.25
  stats = torch.arange(2 * 96, dtype=torch.float32, device=dev).reshape(2, 96)
  ref = stats.clone()
  dist.all_reduce(stats, op=dist.ReduceOp.SUM)
  assert torch.equal(stats, ref)

  group = dist_utils.clone_group(None)
  warm = torch.zeros(1, dtype=torch.float32, device=dev)
  dist.all_reduce(warm, group=group)
  torch.cuda.synchronize(dev)
  arena = torch.randn(3 * 1024 * 1024 + 17, dtype=torch.float32, device=dev)
  ref = arena.clone()
  sync = dist_utils.GradSync(dev, group, bucket_elems=1 << 20)
  # world size 1 skips the collective inside reduce_range; issue the same calls by hand on the
  # side stream so that RCCL sees the bucketed, unaligned slices
  ready = torch.cuda.Event()
  ready.record()
  with torch.cuda.stream(sync.side):
    sync.side.wait_event(ready)
    for o in range(5, arena.numel(), sync.bucket):
      dist.all_reduce(arena[o:min(o + sync.bucket, arena.numel())], op=dist.ReduceOp.SUM,
                      group=group)
  sync.reduce_range(arena, 0, arena.numel())
  sync.finish()
  torch.cuda.synchronize(dev)
  assert torch.equal(arena, ref)
  # bench.py's own helpers on an initialised group
  assert bench._max_over_ranks(0.5, 1, dev) == 0.5
  dist.barrier()
  dist.destroy_process_group()
  print('RCCL_OK')


if __name__ == '__main__':
  main()
