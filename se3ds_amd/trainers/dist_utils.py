"""Replica plumbing shared by the trainer (RCCL on the GPU box, gloo in the CPU tests).
The SE3DS step has exactly three cross-replica exchanges:
  1. SUM of the per-replica, per-tensor-clipped gradients (se3ds_trainer.py:253-257);
  2. SyncBatchNormalization statistics: SUM of [2][C] partial sums, forward and backward;
  3. nothing else -- spectral `u` and EMA are identical on every replica by construction.
"""
import torch
import torch.distributed as dist

GRAD_BUCKET_ELEMS = 64 * 1024 * 1024   # 256 MiB fp32 per all-reduce: large messages for xGMI rings


def world_size(group=None):
  return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def allreduce_arena_sum(arena: torch.Tensor, group=None, bucket_elems: int = GRAD_BUCKET_ELEMS):
  """In-place SUM all-reduce of a flat arena in large contiguous buckets."""
  if world_size(group) == 1:
    return arena
  for o in range(0, arena.numel(), bucket_elems):
    dist.all_reduce(arena[o:o + bucket_elems], op=dist.ReduceOp.SUM, group=group)
  return arena


def shard_batch(batch: dict, rank: int, world: int) -> dict:
  """Rank r takes samples [r*B, (r+1)*B) of the global batch (base_dataset.py:136-143)."""
  out = {}
  for k, v in batch.items():
    n = v.shape[0]
    if n % world != 0:
      raise ValueError(f'global batch {n} of {k} is not divisible by {world} replicas')
    b = n // world
    out[k] = v[rank * b:(rank + 1) * b]
  return out
