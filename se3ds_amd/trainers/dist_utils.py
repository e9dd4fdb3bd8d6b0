"""Replica plumbing shared by the trainer (RCCL on the GPU box, gloo in the CPU tests).
The SE3DS step has exactly three cross-replica exchanges:
  1. SUM of the per-replica, per-tensor-clipped gradients (se3ds_trainer.py:253-257);
  2. SyncBatchNormalization statistics: SUM of [2][C] partial sums, forward and backward;
  3. nothing else -- spectral `u` and EMA are identical on every replica by construction.
"""
import torch
import torch.distributed as dist

GRAD_BUCKET_ELEMS = 64 * 1024 * 1024   # 256 MiB fp32 per all-reduce: large messages for xGMI rings
DRIP_BUCKET_ELEMS = 16 * 1024 * 1024   # 64 MiB: drip-fed behind SyncBN collectives (GradSync)


# Per-step record of the collectives a rank issues (bench.py --gpus N carries its summary on the
# line; tests read it): None = off, else a list of (kind, elements, host time).  Kinds: 'syncbn'
# (one [2][C] statistics sum), 'syncbn_pair' (the two decoders' k-th sums in one all-reduce),
# 'grad_bucket' (a slice of the clipped gradient arena on the side stream), 'grad_arena'.
COLLECTIVE_LOG = None


def log_collective(kind, numel):
  if COLLECTIVE_LOG is not None:
    import time
    COLLECTIVE_LOG.append((kind, int(numel), time.perf_counter()))


# The same step seen from the DEVICE (bench.py --gpus N, one instrumented step): None = off, else a list
# of (kind, start event, end event) recorded on the stream each collective was issued on -- for the
# SyncBN sums that is the compute stream, so their summed durations are time the step was HELD by
# communication; for the gradient buckets it is GradSync's side stream (busy time, hidden unless
# finish() has to wait: 'finish_wait').
COLLECTIVE_EVENTS = None


def all_reduce_sum(t, group, kind):
  """SUM all-reduce of `t` in place + the two logs above."""
  evs = COLLECTIVE_EVENTS
  if evs is not None and t.is_cuda:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    e1.record()
    evs.append((kind, e0, e1))
  else:
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
  log_collective(kind, t.numel())


def summarize_collective_events(evs):
  """{kind: {count, ms}} from COLLECTIVE_EVENTS (call after a device synchronise)."""
  out = {}
  for kind, e0, e1 in evs:
    e = out.setdefault(kind, {'count': 0, 'ms': 0.0})
    e['count'] += 1
    e['ms'] += e0.elapsed_time(e1)
  return out


def summarize_collectives(log):
  """{kind: count, elements, bytes}, totals and the median host-side spacing of the SyncBN sums."""
  out = {}
  for kind, n, _ in log:
    e = out.setdefault(kind, {'count': 0, 'bytes': 0})
    e['count'] += 1
    e['bytes'] += 4 * n
  ts = sorted(t for k, _, t in log if k.startswith('syncbn'))
  gaps = sorted(b - a for a, b in zip(ts, ts[1:]))
  out['total'] = {'count': len(log), 'bytes': sum(4 * n for _, n, _ in log)}
  out['syncbn_host_gap_us_median'] = 1e6 * gaps[len(gaps) // 2] if gaps else None
  # host-side issue timeline: the span from the first to the last collective of the step cut into
  # ten equal parts, collectives per part and kind (forward: singles, then the decoders' pairs;
  # backward: pairs first, gradient buckets leaving while the encoder's singles are still going)
  if log:
    t0, t1 = log[0][2], log[-1][2]
    span = max(t1 - t0, 1e-9)
    tl = {}
    for kind, _, t in log:
      tl.setdefault(kind, [0] * 10)[min(int(10 * (t - t0) / span), 9)] += 1
    out['timeline'] = {'span_ms': 1e3 * (t1 - t0), 'per_tenth': tl}
  return out


def world_size(group=None):
  return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def allreduce_arena_sum(arena: torch.Tensor, group=None, bucket_elems: int = GRAD_BUCKET_ELEMS):
  """In-place SUM all-reduce of a flat arena in large contiguous buckets."""
  if world_size(group) == 1:
    return arena
  for o in range(0, arena.numel(), bucket_elems):
    dist.all_reduce(arena[o:o + bucket_elems], op=dist.ReduceOp.SUM, group=group)
    log_collective('grad_arena', min(bucket_elems, arena.numel() - o))
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


def clone_group(group=None):
  """A second process group (its own RCCL communicator and stream) over the same ranks.
  Collective: every rank of `group` must call it at the same point."""
  ranks = dist.get_process_group_ranks(group if group is not None else dist.group.WORLD)
  return dist.new_group(ranks=ranks)


class GradSync:
  """Overlaps the gradient all-reduce with the backward pass: as soon as a top-level module's
  parameter gradients are final (and clipped, per replica and per tensor, as the reference does
  before aggregation) its slice of the flat gradient arena is all-reduced on a side HIP stream
  while the main stream keeps running backward kernels.  `finish()` joins the streams before
  the optimiser reads the arena.

  PyTorch runs the collectives of one process group FIFO on that group's RCCL stream.  On the
  SHARED communicator (the default) a module's whole slice, handed over at once, would sit in
  front of the next SyncBN sum the backward pass needs and the overlap would be lost; so the slice
  is cut into buckets that are DRIP-FED: one right away, one more behind every SyncBN collective
  of the continuing backward pass (`pump`, wired to `Ctx.after_collective`), the rest at
  `finish()`.  A SyncBN sum then waits for at most one bucket (64 MiB: ~0.3 ms on an 8-GPU xGMI
  ring) that had the preceding layer's backward kernels to overlap with.  With its own
  communicator (`SE3DS_GRAD_SYNC_OWN_COMM=1`) nothing queues behind the buckets and `drip` is
  off: large buckets go out at once."""

  def __init__(self, device, group=None, bucket_elems: int = None, drip: bool = True):
    self.device = torch.device(device)
    self.group = group
    self.drip = drip
    if bucket_elems is None:
      bucket_elems = DRIP_BUCKET_ELEMS if drip else GRAD_BUCKET_ELEMS
    self.bucket = bucket_elems
    self.side = torch.cuda.Stream(device=self.device)
    self.pending = []    # (arena, o, n, ready event) not yet on the side stream
    self.launched = []   # (e0, e1) ranges handed over this step (for tests)
    self.seen = 0            # collectives of this step that called pump() so far
    self.prev_seen = None    # ... of the previous step (None: first step, pace one per call)
    self.last_finish_buckets = 0   # buckets the last finish() still had to issue itself

  def _issue(self, arena, o, n, ready):
    with torch.cuda.stream(self.side):
      self.side.wait_event(ready)
      if world_size(self.group) > 1:
        all_reduce_sum(arena[o:o + n], self.group, 'grad_bucket')

  def reduce_range(self, arena: torch.Tensor, e0: int, e1: int):
    ready = torch.cuda.Event()
    ready.record()   # on the main stream: the slice is clipped and final
    for o in range(e0, e1, self.bucket):
      self.pending.append((arena, o, min(self.bucket, e1 - o), ready))
    self.launched.append((e0, e1))
    self.pump(1 if self.drip else len(self.pending))

  def pump(self, k: int = None):
    """Moves up to k pending buckets onto the side stream.  k=None is the call wired behind
    every SyncBN collective of the backward pass: at least one bucket, and as many as it takes
    to have the queue empty by the LAST collective of the step (their number is known from the
    previous step), so that the bytes in flight follow the remaining backward time and nothing
    but the last segments' own buckets is left for finish()."""
    if k is None:
      self.seen += 1
      k = 1
      if self.prev_seen:
        left = max(self.prev_seen - self.seen + 1, 1)
        k = max(1, -(-len(self.pending) // left))
    while k > 0 and self.pending:
      self._issue(*self.pending.pop(0))
      k -= 1

  def _end_step(self):
    """Pacing bookkeeping of finish() (pure host logic: tests/test_dist_cpu.py drives it without a
    GPU).  A step that paced nothing -- train_d: its only buckets are the discriminator's, handed
    over in one go -- keeps the count of the last step that did, so train_d / train_g_d alternation
    (d_step_per_g_step = 2) does not reset the pace.  When the count CHANGES between two pacing
    steps, the first step after the change runs on the stale figure: more collectives than
    expected -> everything pending goes out behind each one past the old count (drained early);
    fewer -> the tail is left for finish() once, and the next step paces on the new count."""
    self.prev_seen, self.seen = (self.seen or self.prev_seen), 0
    left = len(self.pending)   # what the pacing did NOT hide behind a collective (bench: `finish_buckets`)
    self.pump(len(self.pending))
    return left

  def finish(self):
    self.last_finish_buckets = self._end_step()
    evs = COLLECTIVE_EVENTS
    main = torch.cuda.current_stream(self.device)
    if evs is not None:
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(main)
      main.wait_stream(self.side)
      e1.record(main)
      evs.append(('finish_wait', e0, e1))   # how long Adam waited for the last buckets
    else:
      main.wait_stream(self.side)
    self.launched = []
