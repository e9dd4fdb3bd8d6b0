"""CPU tests (gloo, world_size 2) of the replica plumbing and of the data-parallel semantics the
GPU path implements: SyncBN statistics pooled over replicas, per-replica per-tensor clipping
BEFORE the cross-replica sum, losses pre-divided by the replica count."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nets_torch as O
from se3ds_amd.trainers import dist_utils


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def test_merge_paired_orders_sync_closures_next_to_their_twin():
  from se3ds_amd.hipops import nn
  e = lambda name, tag, sync=0: (name, tag, sync)
  # two branches recorded by two threads in arbitrary interleaving, main-stream entries around
  tape = [e('m0', 0), e('a0', 1), e('b0', 2), e('b1', 2), e('A', 1, 1), e('a1', 1), e('B', 2, 1),
          e('b2', 2), e('a2', 1), e('A2', 1, 1), e('B2', 2, 1), e('b3', 2), e('m1', 0)]
  out = nn._merge_paired(tape)
  names = [x[0] for x in out]
  assert names == ['m0', 'a0', 'b0', 'b1', 'A', 'B', 'a1', 'a2', 'b2', 'A2', 'B2', 'b3', 'm1'], names
  assert [x[2] for x in out if x[0] in ('A', 'B', 'A2', 'B2')] == [2, 2, 2, 2]
  for tag in (1, 2):   # per-branch order untouched
    assert [x[0] for x in out if x[1] == tag] == [x[0] for x in tape if x[1] == tag]
  # unequal numbers of sync closures: left alone (plain, unpaired all-reduces)
  odd = [e('a0', 1), e('A', 1, 1), e('b0', 2)]
  assert nn._merge_paired(odd) == odd


def _toy_params(seed=0):
  from se3ds_amd.models import image_models
  G = image_models.ResNetGenerator(image_size=64, gen_dims=4, z_dim=4, device='cpu', seed=seed)
  D = image_models.SNMultiScaleDiscriminator(dis_dims=4, n_layers=3, device='cpu', seed=seed + 1)
  return ({k: v.clone() for k, v in G.store.views.items()},
          {k: v.clone() for k, v in D.store.views.items()})


def _batch(n, h, seed):
  g = torch.Generator().manual_seed(seed)
  w = 2 * h
  image = torch.rand((n, h, w, 3), generator=g)
  depth = torch.rand((n, h, w, 1), generator=g)
  pm = (torch.rand((n, h, w, 1), generator=g) < 0.5).float()
  bm = torch.zeros((n, h, w, 1))
  bm[:, :h // 8] = 1
  return dict(image=image, depth=depth, proj_mask=pm, proj_image=image * pm,
              proj_depth=depth * pm, blurred_mask=bm)


def _cfg():
  return dict(gen=dict(gen_dims=4, resnet_version='50', context_layer='convs', z_dim=4),
              dis=dict(n_dis=2, n_layers=3, kernel_size=4), lambda_gan=1.0, lambda_kld=10.0,
              lambda_wc=10.0, lambda_depth=100.0, mask_blurred=True,
              g_train=lambda k: not k.endswith(('/u', '/moving_mean', '/moving_variance')),
              d_train=lambda k: not k.endswith('/u'))


def _worker(rank, world, port, mode, out):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  torch.set_num_threads(2)
  try:
    if mode == 'arena':
      arena = torch.arange(1000, dtype=torch.float32) * (rank + 1)
      dist_utils.allreduce_arena_sum(arena, bucket_elems=256)
      out[rank] = arena.numpy().copy()
      full = {'x': torch.arange(8).reshape(4, 2)}
      shard = dist_utils.shard_batch(full, rank, world)
      assert shard['x'].tolist() == [[4 * rank, 4 * rank + 1], [4 * rank + 2, 4 * rank + 3]]
      # gradient traffic runs on its own communicator, interleaved with the default one
      g2 = dist_utils.clone_group(None)
      a = torch.full((300,), float(rank + 1))
      b = torch.full((3,), float(rank + 1))
      dist_utils.allreduce_arena_sum(a, g2, bucket_elems=128)
      dist.all_reduce(b)
      assert a.tolist() == [3.0] * 300 and b.tolist() == [3.0] * 3
    elif mode == 'paired_syncbn':
      # Ctx.run_branches / allreduce_then (se3ds_amd/hipops/nn.py): two structurally identical
      # branches in lockstep share ONE all-reduce per pair of SyncBN sums, forward and backward
      from se3ds_amd.hipops import nn
      ctx = nn.Ctx('cpu', torch.float32, training=True, record=True, world=world)
      seen = {}
      def branch(tag):
        def run():
          vals = []
          for k in range(3):
            t = torch.full((2, 4), float(100 * tag + 10 * k + rank))
            ctx.record(lambda: seen.setdefault('plain', []).append(tag))   # ops between the norms
            ctx.allreduce_sum(t)
            vals.append(t.clone())
            g = torch.full((2, 4), float(1000 * tag + 10 * k + rank))
            def bwd(g=g, k=k, tag=tag):
              ctx.allreduce_then(g, lambda: seen.setdefault('bwd', []).append((tag, k, g.clone())))
            ctx.record(bwd, sync=True)
          return vals
        return run
      res = ctx.run_branches({1: branch(1), 2: branch(2)})
      tri = world * (world - 1) // 2   # (sum of the ranks: every rank contributes value + rank)
      for tag in (1, 2):
        for k, v in enumerate(res[tag]):
          assert torch.equal(v, torch.full((2, 4), float(world * (100 * tag + 10 * k) + tri))), (tag, k)
      assert ctx.collectives == 3, ctx.collectives          # 6 sums, 3 all-reduces
      ctx.backward()
      assert ctx.collectives == 6, ctx.collectives          # ... and 3 more for the 6 backward sums
      got = sorted((tag, k, float(g[0, 0])) for tag, k, g in seen['bwd'])
      assert got == sorted((tag, k, float(world * (1000 * tag + 10 * k) + tri)) for tag in (1, 2)
                           for k in range(3)), got
      # per-branch order of the backward closures is kept: k = 2, 1, 0
      for tag in (1, 2):
        assert [k for t_, k, _ in seen['bwd'] if t_ == tag] == [2, 1, 0]
      out[rank] = np.array([ctx.collectives])
    elif mode == 'replica_step':
      gp, dp = _toy_params()
      full = _batch(2, 64, 5)
      shard = dist_utils.shard_batch(full, rank, world)
      pooled = O.pooled_stats_hook(world)   # SyncBN: differentiable cross-replica sums
      torch.manual_seed(0)
      cfg = _cfg()
      # the oracle's generator takes the hook through a patched Net
      orig = O.Net.__init__
      def patched(self, params, training, stats_hook=None, bn_training=None):
        orig(self, params, training, pooled if training else None, bn_training)
      O.Net.__init__ = patched
      try:
        ref = O.train_g_d(gp, dp, shard, cfg, replicas=world)
      finally:
        O.Net.__init__ = orig
      flat = torch.cat([g.reshape(-1) for g in ref['g_grads'].values()])
      dist_utils.allreduce_arena_sum(flat)   # SUM of per-replica clipped gradients
      out[rank] = dict(sum=flat.numpy().copy(),
                       local_norms=np.array([float(g.norm()) for g in ref['g_grads'].values()]),
                       mm=ref['g_updates']['encoder/bn1/moving_mean'].numpy().copy())
  finally:
    dist.destroy_process_group()


def _run(mode, world=2):
  ctx = mp.get_context('spawn')
  mgr = ctx.Manager()
  out = mgr.dict()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, mode, out)) for r in range(world)]
  for p in procs:
    p.start()
  for p in procs:
    p.join(timeout=600)
    assert p.exitcode == 0
  return dict(out)


def test_arena_allreduce_and_sharding_gloo():
  out = _run('arena')
  expect = np.arange(1000, dtype=np.float32) * 3
  np.testing.assert_array_equal(out[0], expect)
  np.testing.assert_array_equal(out[1], expect)


def test_two_replica_step_semantics_gloo():
  """Two replicas, one sample each: every replica ends with the same summed gradient, the
  clip is applied per replica (local norms <= 5), and the BN moving statistics are identical on
  both replicas because the batch statistics were pooled."""
  out = _run('replica_step')
  np.testing.assert_array_equal(out[0]['sum'], out[1]['sum'])
  np.testing.assert_allclose(out[0]['mm'], out[1]['mm'], rtol=0, atol=0)
  assert out[0]['local_norms'].max() <= 5.0 + 1e-4 and out[1]['local_norms'].max() <= 5.0 + 1e-4
  assert np.isfinite(out[0]['sum']).all()


@pytest.mark.parametrize('world', [2, 4])
def test_paired_syncbn_collectives_gloo(world):
  """the SyncBN sums of two lockstep branches go out as one all-reduce per pair,
  forward (two host threads meeting at a barrier) and backward (twin closures placed next to each
  other): 12 reduced tensors, 6 collectives, every value the cross-replica sum -- on 2 ranks and on 4
  (cfg4 in miniature: the pairing is per rank, the sums are over all of them)."""
  out = _run('paired_syncbn', world)
  assert all(int(out[r][0]) == 6 for r in range(world))


class _PacedSync(dist_utils.GradSync):
  """GradSync's host-side pacing without a GPU: buckets are recorded instead of all-reduced."""

  def __init__(self, bucket):   # pylint: disable=super-init-not-called
    self.group, self.drip, self.bucket = None, True, bucket
    self.pending, self.launched, self.issued = [], [], []
    self.seen, self.prev_seen, self.last_finish_buckets = 0, None, 0

  def _issue(self, arena, o, n, ready):
    self.issued.append((o, n, self.seen))

  def reduce_range(self, arena, e0, e1):   # (the product's, minus the HIP event)
    for o in range(e0, e1, self.bucket):
      self.pending.append((arena, o, min(self.bucket, e1 - o), None))
    self.launched.append((e0, e1))
    self.pump(1 if self.drip else len(self.pending))


def _paced_step(sync, collectives, segments):
  """One train_g_d backward: `collectives` SyncBN sums, a gradient segment of `n` buckets becoming
  ready just before collective k for every (k, n) in `segments`.  Returns (buckets still queued
  right after the LAST collective, buckets finish() had to issue itself)."""
  sync.issued = []
  at = dict(segments)
  queued_after_last = None
  for k in range(collectives):
    if k in at:
      sync.reduce_range(None, 0, at[k] * sync.bucket)
    sync.pump()   # Ctx.after_collective
    queued_after_last = len(sync.pending)
  left = sync._end_step()
  assert not sync.pending
  return queued_after_last, left


def test_drip_pacing_drains_by_the_last_collective_and_follows_a_changing_count():
  """GradSync.pump paces the drip-fed gradient buckets so that the queue is
  empty by the LAST SyncBN collective of the backward pass, whose number it only knows from the
  previous step.  Steady state; train_d steps in between (they pace nothing and must not reset the
  count); the count GROWING (everything queued goes out behind each extra collective) and
  SHRINKING (one step leaves a tail to finish(), the next one is paced on the new count)."""
  sync = _PacedSync(bucket=16)
  segs = [(0, 40), (50, 30), (120, 25)]   # 95 buckets over a 194-collective backward pass
  # first step: no history, one bucket per collective -> 194 >= 95: drained
  q, left = _paced_step(sync, 194, segs)
  assert (q, left) == (0, 0) and sync.prev_seen == 194
  # steady state: drained by the last collective, and not all up front (the bytes in flight follow
  # the remaining backward time: the first segment's 40 buckets are spread over many collectives)
  q, left = _paced_step(sync, 194, segs)
  assert (q, left) == (0, 0)
  first_seg = [s for o, n, s in sync.issued[:40]]
  assert max(first_seg) - min(first_seg) >= 20, first_seg
  # a train_d step in between: reduce_range + finish only, no pacing calls
  sync.reduce_range(None, 0, 5 * sync.bucket)
  assert sync._end_step() == 4 and sync.prev_seen == 194
  q, left = _paced_step(sync, 194, segs)
  assert (q, left) == (0, 0)
  # the count grows (e.g. another model width): past the stale count every call empties the queue
  q, left = _paced_step(sync, 260, segs + [(230, 12)])
  assert (q, left) == (0, 0) and sync.prev_seen == 260
  # the count shrinks: the stale figure paces too slowly ONCE ...
  late = [(0, 40), (80, 50)]
  q1, left1 = _paced_step(sync, 100, late)
  assert left1 == q1 and left1 > 0 and sync.prev_seen == 100
  # ... and the next step with the same shape is drained by its last collective again
  q2, left2 = _paced_step(sync, 100, late)
  assert (q2, left2) == (0, 0)
  # a segment that only becomes ready at the very last collective is all finish() ever has to send
  q3, left3 = _paced_step(sync, 100, [(0, 40), (99, 3)])
  assert (q3, left3) == (0, 0)
