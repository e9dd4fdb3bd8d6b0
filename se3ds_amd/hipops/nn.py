"""Differentiable building blocks of the SE3DS G/D step on libse3ds_hip.so.

Everything numeric happens in our HIP kernels.  torch is used for device memory
(`torch.empty`, views), streams and `torch.distributed` only.  Reverse mode is a small tape:
each op appends a closure; gradient accumulation for tensors with several consumers runs in
`se3ds_add`, parameter gradients are written straight into a flat fp32 arena.
"""
import math
import threading
from typing import List, Optional

import numpy as np
import os

import torch
import torch.distributed as dist

from se3ds_amd import _lib
from se3ds_amd import hipops  # noqa: F401  (registers the ctypes signatures)

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
BN_EPS, BN_MOMENTUM = 1e-3, 0.99   # Keras SyncBatchNormalization defaults
IN_EPS = 1e-3                      # tfa InstanceNormalization default


def _L():
  return _lib.lib()


def _log_collective(kind, numel):
  from se3ds_amd.trainers import dist_utils   # (late: dist_utils does not import this module)
  dist_utils.log_collective(kind, numel)


def _all_reduce_sum(t, group, kind):
  from se3ds_amd.trainers import dist_utils
  dist_utils.all_reduce_sum(t, group, kind)


def _chk(rc, what):
  _lib.check(rc, what)


# ------------------------------------------------------------------------------ parameters
class ParamStore:
  """Flat fp32 arenas: `theta` (trainable), `state` (non-trainable: BN moving statistics,
  spectral `u`), plus matching gradient / Adam / EMA arenas created on demand."""

  def __init__(self):
    self._specs = []   # (name, shape, init, trainable)
    self.theta = None
    self.state = None
    self._grad = None
    self.views = {}
    self.grad_views = _LazyGradViews(self)
    self.offsets = {}
    self.trainable_names = []
    self.state_names = []
    self.version = 0   # bumped whenever theta changes (compute-dtype copies are refreshed)

  def add(self, name, shape, init, trainable=True):
    assert self.theta is None, 'store already finalized'
    assert name not in [s[0] for s in self._specs], name
    self._specs.append((name, tuple(int(s) for s in shape), init, trainable))
    return name

  def finalize(self, device, generator: Optional[torch.Generator] = None):
    tr = [s for s in self._specs if s[3]]
    st = [s for s in self._specs if not s[3]]
    def pack(specs):
      off, offs = 0, {}
      for name, shape, _, _ in specs:
        offs[name] = (off, int(np.prod(shape)) if len(shape) else 1, shape)
        off += offs[name][1]
        off = (off + 3) // 4 * 4   # keep every tensor 16-byte aligned
      return off, offs
    n_tr, off_tr = pack(tr)
    n_st, off_st = pack(st)
    # Initial values are drawn where the generator lives: a CPU generator (default, bit-stable
    # across machines) or a device generator (1.1 B parameters in seconds instead of minutes).
    init_dev = generator.device if generator is not None else torch.device('cpu')
    host_tr = torch.zeros(max(n_tr, 4), dtype=torch.float32, device=init_dev)
    host_st = torch.zeros(max(n_st, 4), dtype=torch.float32, device=init_dev)
    for specs, host, offs in ((tr, host_tr, off_tr), (st, host_st, off_st)):
      for name, shape, init, _ in specs:
        o, n, _ = offs[name]
        host[o:o + n] = init(shape, generator).reshape(-1)
    self.theta = host_tr.to(device)
    self.state = host_st.to(device)
    self._grad = None
    for name, (o, n, shape) in off_tr.items():
      self.views[name] = self.theta[o:o + n].view(shape)
      self.offsets[name] = ('theta', o, n)
    for name, (o, n, shape) in off_st.items():
      self.views[name] = self.state[o:o + n].view(shape)
      self.offsets[name] = ('state', o, n)
    self.trainable_names = [s[0] for s in tr]
    self.state_names = [s[0] for s in st]
    self._off_tr = off_tr
    return self

  def __getitem__(self, name):
    return self.views[name]

  @property
  def grad(self):
    """Gradient arena, allocated on first use (the EMA generator never needs one)."""
    if self._grad is None:
      self._grad = torch.zeros_like(self.theta)
      for name, (o, n, shape) in self._off_tr.items():
        self.grad_views[name] = self._grad[o:o + n].view(shape)
    return self._grad

  def load_dict(self, d):
    """Overwrite parameters from {name: array-like} (tests inject identical weights)."""
    for k, v in d.items():
      t = torch.as_tensor(np.asarray(v), dtype=torch.float32).to(self.theta.device)
      self.views[k].copy_(t.reshape(self.views[k].shape))
    self.version += 1

  def to_dict(self):
    return {k: v.detach().cpu().numpy().copy() for k, v in self.views.items()}

  def segments(self, groups):
    """{name: (t0, t1, e0, e1)}: tensor-index and element ranges of the trainable tensors whose
    names start with one of the group's prefixes + '/' (registered contiguously).  `groups` is
    {name: (prefix, ...)} or a sequence of prefixes (each its own group)."""
    if not isinstance(groups, dict):
      groups = {g: (g,) for g in groups}
    out = {}
    for name, prefixes in groups.items():
      idx = [i for i, n in enumerate(self.trainable_names)
             if any(n.startswith(pre + '/') for pre in prefixes)]
      if not idx:
        continue
      assert idx == list(range(idx[0], idx[-1] + 1)), f'{name} is not contiguous'
      o0 = self._off_tr[self.trainable_names[idx[0]]][0]
      last = self._off_tr[self.trainable_names[idx[-1]]]
      out[name] = (idx[0], idx[-1] + 1, o0, last[0] + last[1])
    return out

  def chunk_tables(self, chunk=65536):
    """(chunks int64 [nchunks,3], tensor_chunk_start int64 [T+1]) for the multi-tensor ops."""
    chunks, starts = [], [0]
    for t, name in enumerate(self.trainable_names):
      o, n, _ = self._off_tr[name]
      for s in range(0, n, chunk):
        chunks.append((t, o + s, min(chunk, n - s)))
      starts.append(len(chunks))
    dev = self.theta.device
    return (torch.tensor(chunks, dtype=torch.int64, device=dev).reshape(-1, 3).contiguous(),
            torch.tensor(starts, dtype=torch.int64, device=dev))


class _LazyGradViews(dict):
  """name -> view into the gradient arena; touching it allocates the arena."""

  def __init__(self, store):
    super().__init__()
    self._store = store

  def __missing__(self, name):
    self._store.grad  # allocates and fills every view
    return dict.__getitem__(self, name)


def _gdev(gen):
  return gen.device if gen is not None else torch.device('cpu')


def glorot_uniform(shape, gen):
  rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
  fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
  limit = math.sqrt(6.0 / (fan_in + fan_out))
  return (torch.rand(shape, generator=gen, dtype=torch.float32, device=_gdev(gen)) * 2 - 1) * limit


def zeros_init(shape, gen):
  return torch.zeros(shape, dtype=torch.float32, device=_gdev(gen))


def ones_init(shape, gen):
  return torch.ones(shape, dtype=torch.float32, device=_gdev(gen))


def truncated_normal_init(shape, gen, std=0.05):
  # tf.initializers.TruncatedNormal(): N(0, 0.05) redrawn outside 2 sigma
  dev = _gdev(gen)
  t = torch.randn(shape, generator=gen, dtype=torch.float32, device=dev)
  for _ in range(8):
    bad = t.abs() > 2
    if not bool(bad.any()):
      break
    t = torch.where(bad, torch.randn(shape, generator=gen, dtype=torch.float32, device=dev), t)
  return t.clamp(-2, 2) * std


# --------------------------------------------------------------------------------- context
class Var:
  """Activation handle: NHWC tensor in the compute dtype plus its (lazy) gradient."""
  __slots__ = ('data', '_grad', 'gver', 'requires_grad', 'col_stats', 'grad_pre_act', 'shared',
               'grad_ev', 'bn_src', 'bn_stats', 'row_sink', 'grad_scaled')

  def __init__(self, data, requires_grad=True):
    self.data = data
    self._grad = None
    self.gver = 0           # bumped by every write to .grad (assignment or in-place accumulation)
    self.bn_src = None      # batch-norm outputs: (x, act mask, mean, rstd, act, alpha) of the norm
    self.bn_stats = None    # ... (stats rows, gver): backward statistics a data-gradient epilogue
                            # took from the gradient it stored (valid while gver is unchanged)
    self.row_sink = None    # outputs of biased partial convs: (out row scale, bias row scale,
                            # store, bias name) -- a batch norm that is the only consumer stores
                            # its dx pre-scaled and writes the bias gradient (norm_bwd_apply_rows)
    self.grad_scaled = None   # ... gver at which it did
    self.requires_grad = requires_grad
    self.col_stats = None   # conv outputs: partial column sums for a following batch norm
    self.grad_pre_act = False   # the consumer already applied this tensor's activation derivative
    self.shared = False     # consumed by branches that run on different HIP streams (Ctx.streams)
    self.grad_ev = None     # ... then: event behind the last write of .grad

  @property
  def shape(self):
    return self.data.shape

  @property
  def grad(self):
    return self._grad

  @grad.setter
  def grad(self, g):
    self._grad = g
    self.gver += 1


class Ctx:
  """Per-call execution context: compute dtype, training flag, tape, replica group."""

  def __init__(self, device, dtype=torch.float32, training=False, record=False,
               group=None, world=None):
    self.device = torch.device(device)
    self.dtype = dtype
    self.code = _lib.dtype_code(torch.empty(0, dtype=dtype))
    self.training = training
    self.tape = [] if record else None
    self.param_grads = True   # False: backward passes only propagate input gradients
    # PartialConv masks hold only 0/1 (true for the reference's data pipeline and for every
    # update_mask derived from a binary mask).  Set False for fractional masks: exact, slower.
    self.binary_masks = True
    self.on_segment = None    # callback(name): a top-level module's parameter gradients are final
    self.batch_limit = None   # backward passes that only concern the first samples of the batch
    self.act_taps = None      # tests: dict layer name -> activated output (sign decisions)
    # Independent branches of the graph (the generator's two decoders) on their own HIP streams:
    # {tag: torch.cuda.Stream}, None = everything on the current stream.  Ops record the branch
    # they ran in; backward() replays each branch's closures on its stream (see branch()).
    self.streams = None
    self.stream_phases = ('fwd', 'bwd')   # debugging: restrict the branch streams to one pass
    # Weight gradients leave the backward pass's dependency chain (dgrad -> norm backward ->
    # dgrad ...): with a stream here, every conv layer's wgrad goes to it behind an event on dy,
    # and MFMA-bound wgrads run under the HBM-bound normalisation kernels of the chain.
    self.wgrad_stream = None
    # Deferred split reductions of the weight gradients (round 4): a list collects one row per conv
    # layer [slabs, splits, n / 4, destination]; flush_wgrad_reduces() runs them as ONE launch (the
    # trainer: per module, on the optimiser's side stream).  None: every layer reduces right away.
    self.wgrad_defer = None
    self._tl = threading.local()   # branch_tag is per host thread (paired branches run in two)
    self._forked = set()
    # Several replicas: two structurally identical branches run in lockstep so that their k-th
    # SyncBatchNormalization sums share ONE all-reduce (run_branches / _PairSync forward,
    # allreduce_then backward).  collectives counts the all-reduces issued through this context.
    self.pair = None
    self.collectives = 0
    self._bwd_pending = None
    self._cur_paired = False
    self.group = group
    # Replica count of the STRATEGY that built the model, never probed from global
    # torch.distributed state: a one-device model inside an initialised process group must not
    # all-reduce its batch-norm sums.  world=None with a group means "that group's size".
    if world is None:
      world = dist.get_world_size(group) if group is not None else 1
    self.world = int(world)
    if group is not None and self.world != dist.get_world_size(group):
      raise ValueError(f'Ctx: world={self.world} does not match the replica group '
                       f'({dist.get_world_size(group)} ranks)')
    if group is None and self.world > 1 and not (dist.is_initialized() and
                                                 dist.get_world_size() == self.world):
      raise ValueError(f'Ctx: world={self.world} without a group needs a default process group '
                       'of that size')
    self._ws = {}

  def ws(self, key, nbytes):
    if self.streams is not None:   # scratch is per stream: branches run concurrently
      key = (key, torch.cuda.current_stream(self.device).cuda_stream)
    elif self.pair is not None:    # ... or per branch: two host threads issue them in lockstep
      key = (key, self.branch_tag)
    t = self._ws.get(key)
    if t is None or t.numel() < nbytes:
      t = torch.empty((int(nbytes),), dtype=torch.uint8, device=self.device)
      self._ws[key] = t
    return t

  def empty(self, shape, dtype=None):
    return torch.empty(shape, dtype=dtype or self.dtype, device=self.device)

  @property
  def branch_tag(self):
    return getattr(self._tl, 'tag', 0)

  @branch_tag.setter
  def branch_tag(self, tag):
    self._tl.tag = tag

  def record(self, fn, sync=False):
    """sync: the closure issues one cross-replica all-reduce through allreduce_then."""
    if self.tape is not None:
      self.tape.append((fn, self.branch_tag, 1 if sync else 0))

  def run_branches(self, fns):
    """fns = {tag: callable}: independent branches of the graph; returns {tag: result}.  One
    replica: one after the other, each under branch(tag) (its own HIP stream when the context has
    streams).  Several replicas and exactly two branches: in two host threads in LOCKSTEP -- both
    reach their k-th SyncBatchNormalization before either continues, and the two [2][C] sums go
    out as one all-reduce (the two decoders hold 168 of the generator's 279 batch
    norms).  The kernels of both threads go to the current stream; what a branch touches is its
    own (layers, activations, per-branch scratch)."""
    if self.world > 1 and len(fns) == 2:
      return _run_paired(self, fns)
    out = {}
    keep = self.streams
    if 'fwd' not in self.stream_phases:
      self.streams = None
    try:
      for tag, fn in fns.items():
        with self.branch(tag):
          out[tag] = fn()
      self.join()
    finally:
      self.streams = keep
    return out

  def branch(self, tag):
    """with ctx.branch(tag): ops of one independent branch.  With Ctx.streams the branch's
    kernels go to streams[tag], which first waits for everything issued on the main stream (fork);
    join() makes the main stream wait for the branches.  While branches are open nothing may be
    issued on the main stream.  Why: the two decoders are ~90 % of the generator and independent,
    so the tail of one's kernel overlaps the ramp of the other's, and the HBM-bound normalisation
    kernels of one run under the MFMA-bound convolutions of the other."""
    return _Branch(self, tag)

  def join(self):
    if self._forked and self.streams is not None:
      main = torch.cuda.current_stream(self.device)
      for tag in sorted(self._forked):
        main.wait_stream(self.streams[tag])
      self._forked.clear()

  def on_wgrad_stream(self, *reads):
    """with ctx.on_wgrad_stream(dy, ...): a weight-gradient launch.  Returns a context manager
    that switches to the wgrad stream behind an event on the current stream; `reads` are tensors
    of the current stream's allocator that the launch reads (they may be freed by the caller
    right away: record_stream keeps their memory until the wgrad stream has passed)."""
    return _WgradScope(self, reads)

  def wgrad_event(self):
    """Event behind everything issued to the wgrad stream so far (None without one)."""
    if self.wgrad_stream is None:
      return None
    ev = torch.cuda.Event()
    ev.record(self.wgrad_stream)
    return ev

  def backward(self):
    tape, self.tape = self.tape, []
    if self.world > 1:
      tape = _merge_paired(tape)
    keep = self.streams
    if 'bwd' not in self.stream_phases:
      self.streams = None
    try:
      self._replay(tape)
    finally:
      self.streams = keep
    if self.wgrad_stream is not None:   # every gradient is final when backward() returns
      torch.cuda.current_stream(self.device).wait_stream(self.wgrad_stream)

  def _replay(self, tape):
    cur, scope = 0, None
    try:
      for fn, tag, sync in reversed(tape):
        self._cur_paired = sync == 2
        t = tag if self.streams is not None else 0
        if t != cur:   # (one stream switch per run of closures, not per closure)
          # deferred weight-gradient reductions never cross a stream switch: what is pending was
          # issued on the stream we are leaving
          flush_wgrad_reduces(self)
          if scope is not None:
            scope.__exit__(None, None, None)
            scope = None
          if t == 0:
            self.join()
          else:
            scope = self.branch(t)
            scope.__enter__()
          cur = t
        fn()
      flush_wgrad_reduces(self)
    finally:
      # (also on an exception out of a closure: torch's current stream must not stay on a branch
      # stream, and a half-paired SyncBN backward must not leak into the next pass)
      if scope is not None:
        scope.__exit__(None, None, None)
      self._cur_paired = False
      pending, self._bwd_pending = self._bwd_pending, None
      self.join()
    assert pending is None, 'unpaired SyncBN backward'

  def mark_segment(self, name):
    """Call BEFORE running a top-level module in the forward pass: in the backward pass the
    marker fires after every op of that module, i.e. when its parameter gradients are final."""
    if self.tape is not None:
      def fire():
        if self.on_segment is not None and self.param_grads:
          self.on_segment(name)
      self.tape.append((fire, self.branch_tag, 0))

  def _allreduce_joint(self, a, b):
    """ONE all-reduce for two tensors (a pair of SyncBN sums)."""
    buf = torch.cat([a.reshape(-1), b.reshape(-1)])
    _all_reduce_sum(buf, self.group, 'syncbn_pair')
    a.copy_(buf[:a.numel()].view_as(a))
    b.copy_(buf[a.numel():].view_as(b))
    self._after_collective()

  def _after_collective(self):
    self.collectives += 1
    cb = getattr(self, 'after_collective', None)
    if cb is not None:
      cb()   # (GradSync.pump: pending gradient buckets go out behind this collective)

  def allreduce_sum(self, t):
    if self.world > 1:
      pair = self.pair
      if pair is not None and self.branch_tag:
        pair.allreduce(t, self.branch_tag)   # forward of two branches in lockstep threads
        return
      _all_reduce_sum(t, self.group, 'syncbn')
      self._after_collective()

  def allreduce_then(self, t, cont):
    """Backward closures: all-reduce `t`, then run cont().  When backward() has placed this
    closure next to its twin of the other branch (_merge_paired), the first of the two only parks
    (t, cont); the second reduces both tensors in one collective and runs both continuations."""
    if self.world > 1:
      if self._cur_paired:
        on_streams = self.streams is not None and t.is_cuda
        if self._bwd_pending is None:
          ev = None
          if on_streams:
            ev = torch.cuda.Event()
            ev.record()   # behind this branch's statistics kernel, on its stream
          self._bwd_pending = (t, cont, ev, self.branch_tag)
          return
        t0, cont0, ev0, tag0 = self._bwd_pending
        self._bwd_pending = None
        if ev0 is not None:
          torch.cuda.current_stream(self.device).wait_event(ev0)
        self._allreduce_joint(t0, t)
        if ev0 is not None and tag0 != self.branch_tag:
          # the twin's continuation runs on ITS stream, behind the copy-back issued on this one
          ev1 = torch.cuda.Event()
          ev1.record()
          with self.branch(tag0):
            torch.cuda.current_stream(self.device).wait_event(ev1)
            cont0()
        else:
          cont0()
        cont()
        return
      self.allreduce_sum(t)
    cont()


class _WgradScope:
  def __init__(self, ctx, reads):
    self.ctx, self.reads, self.scope = ctx, reads, None

  def __enter__(self):
    ws = self.ctx.wgrad_stream
    if ws is not None:
      ev = torch.cuda.Event()
      ev.record()
      ws.wait_event(ev)
      for t in self.reads:
        if t is not None:
          t.record_stream(ws)
      self.scope = torch.cuda.stream(ws)
      self.scope.__enter__()
    return self

  def __exit__(self, *a):
    if self.scope is not None:
      self.scope.__exit__(*a)


class _Branch:
  def __init__(self, ctx, tag):
    self.ctx, self.tag, self.scope, self.prev = ctx, tag, None, 0

  def __enter__(self):
    ctx = self.ctx
    self.prev, ctx.branch_tag = ctx.branch_tag, self.tag
    if ctx.streams is not None and self.tag:
      s = ctx.streams[self.tag]
      if self.tag not in ctx._forked:
        s.wait_stream(torch.cuda.current_stream(ctx.device))
        ctx._forked.add(self.tag)
      self.scope = torch.cuda.stream(s)
      self.scope.__enter__()
    return self

  def __exit__(self, *a):
    if self.scope is not None:
      self.scope.__exit__(*a)
    self.ctx.branch_tag = self.prev


class _PairSync:
  """Rendezvous of two branch threads at their SyncBN sums (forward pass).  With per-branch HIP
  streams (Ctx.streams, round 4) the two sums live on two streams: the partner records an event
  behind its statistics kernel, the leader's stream waits for it, reduces both tensors in one
  collective, and the partner's stream waits for the event behind the copy-back."""

  def __init__(self, ctx, lead, other):
    self.ctx, self.lead, self.other = ctx, lead, other
    self.barrier = threading.Barrier(2)
    self.slot = {}
    self.ev = {}

  def allreduce(self, t, tag):
    ctx = self.ctx
    streams = ctx.streams is not None and t.is_cuda
    self.slot[tag] = t
    if streams and tag != self.lead:
      e = torch.cuda.Event()
      e.record()            # (this thread's current stream = its branch stream)
      self.ev['in'] = e
    self.barrier.wait()
    if tag == self.lead:
      if streams:
        torch.cuda.current_stream(ctx.device).wait_event(self.ev['in'])
      # both statistics kernels are enqueued (and, across streams, ordered by the event)
      ctx._allreduce_joint(self.slot[self.lead], self.slot[self.other])
      if streams:
        e = torch.cuda.Event()
        e.record()
        self.ev['out'] = e
    self.barrier.wait()   # the partner continues only behind the enqueued copy-back
    if streams and tag != self.lead:
      torch.cuda.current_stream(ctx.device).wait_event(self.ev['out'])


def _run_paired(ctx, fns):
  lead, other = sorted(fns)
  pair = _PairSync(ctx, lead, other)
  res, err = {}, []
  on_gpu = ctx.device.type == 'cuda'
  dev, stream = ctx.device, (torch.cuda.current_stream(ctx.device) if on_gpu else None)

  keep = ctx.streams
  if 'fwd' not in ctx.stream_phases:
    ctx.streams = None   # (debugging switch: the forward pass on one stream, as run_branches does)

  def work(tag):
    try:
      if on_gpu:
        torch.cuda.set_device(dev)
        with torch.cuda.stream(stream):
          with ctx.branch(tag):
            res[tag] = fns[tag]()
      else:   # (host-only contexts: the CPU tests of the pairing logic)
        with ctx.branch(tag):
          res[tag] = fns[tag]()
    except BaseException as e:   # noqa: B902  (re-raised in the caller's thread)
      err.append(e)
      pair.barrier.abort()

  ctx.pair = pair
  try:
    th = threading.Thread(target=work, args=(other,), name='se3ds-branch')
    th.start()
    work(lead)
    th.join()
  finally:
    ctx.pair = None
    ctx.streams = keep
  if err:
    real = [e for e in err if not isinstance(e, threading.BrokenBarrierError)]
    raise (real or err)[0]
  return res


def _merge_paired(tape):
  """Reorders every run of branch-tagged tape entries (two branches, recorded by two threads in
  arbitrary interleaving) so that the k-th sync closure of one branch sits NEXT to the k-th of
  the other: a_seg0, b_seg0, a_sync0, b_sync0, a_seg1, ...  Per-branch order is kept (the
  branches are independent), and the adjacent sync closures are flagged 2 = paired."""
  out, i, n = [], 0, len(tape)
  while i < n:
    if tape[i][1] == 0:
      out.append(tape[i])
      i += 1
      continue
    j = i
    while j < n and tape[j][1] != 0:
      j += 1
    run = tape[i:j]
    tags = sorted({e[1] for e in run})
    merged = None
    if len(tags) == 2:
      parts = {t: [e for e in run if e[1] == t] for t in tags}
      if sum(e[2] for e in parts[tags[0]]) == sum(e[2] for e in parts[tags[1]]):
        def segments(entries):
          segs, cur = [], []
          for e in entries:
            if e[2]:
              segs.append((cur, e))
              cur = []
            else:
              cur.append(e)
          return segs, cur
        sa, ta = segments(parts[tags[0]])
        sb, tb = segments(parts[tags[1]])
        merged = []
        for (seg_a, sync_a), (seg_b, sync_b) in zip(sa, sb):
          merged += seg_a + seg_b + [(sync_a[0], sync_a[1], 2), (sync_b[0], sync_b[1], 2)]
        merged += ta + tb
    if merged is None:
      # asymmetric branches (different SyncBN counts, or not exactly two tags): no pairing, and NOT
      # the recorded order either -- that is the host threads' interleaving and differs from rank
      # to rank, so equal-shaped all-reduces of different layers could meet.  One branch after the
      # other, by tag: the same order on every rank.
      merged = [e for t in tags for e in run if e[1] == t]
    out += merged
    i = j
  return out


def _grad_wait(var):
  """Before reading or modifying the gradient of a Var that several stream branches feed."""
  if var.grad_ev is not None:
    torch.cuda.current_stream().wait_event(var.grad_ev)


def _grad_mark(var):
  if var.shared:
    var.grad_ev = torch.cuda.Event()
    var.grad_ev.record()


_WS = {}


def make_stream(device, role: str):
  """A side stream of the step's schedule (`role`: 'branch1' / 'branch2' = the two decoders,
  'optimizer', 'discriminator', 'wgrad').  SE3DS_CU_MASK (round 6, VERDICT r5 item 4 -- the measured
  attempt at CO-RUNNING the HBM-bound kernels under the convolutions: an 8-wave conv workgroup at 256
  registers per lane owns a CU's whole register file and 150 KB of its LDS, so a kernel of another
  stream only ever gets CUs between conv workgroups; a CU mask gives it CUs of its own) selects a
  plan of `hipExtStreamCreateWithCUMask` masks:
    halves     branch1 on CUs 0-127, branch2 on 128-255 (contiguous in the mask's enumeration)
    alternate  branch1 on the even, branch2 on the odd CUs
    opt32 / opt64   only the optimizer stream is confined, to the last 32 / 64 CUs
  Unset (the default, and what every measurement outside tools/probes/cu_mask_ab.sh runs): plain
  streams.  DESIGN.md section 3.4 has the A/B."""
  plan = os.environ.get('SE3DS_CU_MASK')
  dev = torch.device(device)
  mask = None
  if plan and dev.type == 'cuda':
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    bits = None
    if plan == 'halves' and role in ('branch1', 'branch2'):
      bits = [(i < ncu // 2) == (role == 'branch1') for i in range(ncu)]
    elif plan == 'alternate' and role in ('branch1', 'branch2'):
      bits = [(i % 2 == 0) == (role == 'branch1') for i in range(ncu)]
    elif plan in ('opt32', 'opt64') and role == 'optimizer':
      k = int(plan[3:])
      bits = [i >= ncu - k for i in range(ncu)]
    if bits is not None:
      words = (ncu + 31) // 32
      mask = [0] * words
      for i, b in enumerate(bits):
        if b:
          mask[i // 32] |= 1 << (i % 32)
  if mask is None:
    return torch.cuda.Stream(dev)
  import ctypes
  hip = ctypes.CDLL('libamdhip64.so')
  st = ctypes.c_void_p()
  arr = (ctypes.c_uint32 * len(mask))(*mask)
  with torch.cuda.device(dev):
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(mask)), arr)
  if rc != 0:
    raise _lib.Se3dsHipError(f'hipExtStreamCreateWithCUMask failed ({rc}) for SE3DS_CU_MASK={plan}')
  return torch.cuda.ExternalStream(st.value, device=dev)


class ConvProfiler:
  """HIP-event timing of every convolution launch (bench.py roofline leg).  Events are
  recorded on the stream the kernels are launched on (torch's current stream)."""

  def __init__(self):
    self.records = []   # (kind, flops, start_event, end_event)

  def start(self):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e

  def stop(self, kind, flops, e0, tag=None):
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    self.records.append((kind, flops, e0, e1, tag))

  def by_shape(self):
    torch.cuda.synchronize()
    agg = {}
    for kind, fl, e0, e1, tag in self.records:
      a = agg.setdefault((kind, tag), [0.0, 0.0, 0])
      a[0] += e0.elapsed_time(e1); a[1] += fl; a[2] += 1
    return agg

  def summary(self):
    torch.cuda.synchronize()
    tot_ms = tot_fl = 0.0
    by = {}
    for kind, fl, e0, e1, _ in self.records:
      ms = e0.elapsed_time(e1)
      tot_ms += ms
      tot_fl += fl
      b = by.setdefault(kind, [0.0, 0.0, 0])
      b[0] += ms; b[1] += fl; b[2] += 1
    return dict(ms=tot_ms, flops=tot_fl, launches=len(self.records),
                by_kind={k: dict(ms=v[0], tflop=v[1] / 1e12, launches=v[2],
                                 tflops=(v[1] / (v[0] * 1e-3) / 1e12) if v[0] > 0 else 0.0)
                         for k, v in by.items()})


_PROF = None


def set_conv_profiler(p):
  global _PROF
  _PROF = p


def conv_profiler():
  return _PROF


class _Timed:
  """with _Timed(kind, flops): launch  -- no-op unless a profiler is installed."""

  def __init__(self, kind, flops, tag=None):
    self.kind, self.flops, self.tag = kind, flops, tag

  def __enter__(self):
    self.e0 = _PROF.start() if _PROF is not None else None

  def __exit__(self, *a):
    if _PROF is not None and self.e0 is not None:
      _PROF.stop(self.kind, self.flops, self.e0, self.tag)


def _global_ws(device, key, nbytes):
  k = (str(device), key, torch.cuda.current_stream(device).cuda_stream)
  t = _WS.get(k)
  if t is None or t.numel() < nbytes:
    t = torch.empty((int(nbytes),), dtype=torch.uint8, device=device)
    _WS[k] = t
  return t


def accumulate(var: Var, g: torch.Tensor):
  """var.grad += g with our own kernel (tensors with several consumers)."""
  if not var.requires_grad:
    return
  if var.grad is None:
    var.grad = g
  else:
    _grad_wait(var)
    _chk(_L().se3ds_add(var.grad.data_ptr(), g.data_ptr(), _lib.dtype_code(g), g.numel(),
                        var.grad.data_ptr(), _lib.stream()), 'se3ds_add')
    var.gver += 1
  _grad_mark(var)


def to_var(ctx: Ctx, t: torch.Tensor, requires_grad=False) -> Var:
  """fp32 NHWC API tensor -> compute-dtype activation."""
  out = ctx.empty(t.shape)
  c = t.shape[-1]
  rows = t.numel() // c
  t = t.contiguous()
  _chk(_L().se3ds_copy_channels(t.data_ptr(), _lib.dtype_code(t), c, 0, out.data_ptr(), ctx.code,
                                c, 0, c, rows, _lib.stream()), 'se3ds_copy_channels')
  return Var(out, requires_grad)


def concat_channels(ctx: Ctx, parts: List, out_dtype=None) -> torch.Tensor:
  """tf.concat(axis=-1) of NHWC tensors (any of fp32 / compute dtype) into one tensor."""
  out_dtype = out_dtype or ctx.dtype
  lead = parts[0].shape[:-1]
  ctot = sum(p.shape[-1] for p in parts)
  out = torch.empty(tuple(lead) + (ctot,), dtype=out_dtype, device=ctx.device)
  rows = out.numel() // ctot
  c0 = 0
  for p in parts:
    p = p.contiguous()
    c = p.shape[-1]
    _chk(_L().se3ds_copy_channels(p.data_ptr(), _lib.dtype_code(p), c, 0, out.data_ptr(),
                                  _lib.dtype_code(out), ctot, c0, c, rows, _lib.stream()),
         'se3ds_copy_channels')
    c0 += c
  return out


def slice_channels(t: torch.Tensor, c0: int, n: int, out_dtype) -> torch.Tensor:
  c = t.shape[-1]
  out = torch.empty(tuple(t.shape[:-1]) + (n,), dtype=out_dtype, device=t.device)
  rows = t.numel() // c
  _chk(_L().se3ds_copy_channels(t.data_ptr(), _lib.dtype_code(t), c, c0, out.data_ptr(),
                                _lib.dtype_code(out), n, 0, n, rows, _lib.stream()),
       'se3ds_copy_channels')
  return out


# ------------------------------------------------------------------------------ conv layers
def conv_out_size(size, k, stride, padding, pad):
  """(out, pad_before) for an explicit PadLayer `pad` followed by a VALID / SAME conv."""
  if padding == 'VALID':
    return (size + 2 * pad - k) // stride + 1, pad
  assert pad == 0
  out = -(-size // stride)
  total = max((out - 1) * stride + k - size, 0)
  return out, total // 2


class ConvLayer:
  """Parameters + operand copies of one convolution (plain / partial / spectral)."""

  def __init__(self, store: ParamStore, name, cin, cout, k, stride=1, padding='VALID',
               use_bias=True, kind='plain', transpose=False):
    # kind: 'plain' | 'partial' | 'spectral' | 'partial_spectral'
    self.store, self.name = store, name
    self.cin, self.cout, self.k, self.stride, self.padding = cin, cout, k, stride, padding.upper()
    self.kind, self.transpose = kind, transpose
    # Keras layouts: Conv2D (kh,kw,cin,cout); Conv2DTranspose (kh,kw,cout_T,cin_T)
    kshape = (k, k, cout, cin) if transpose else (k, k, cin, cout)
    store.add(name + '/kernel', kshape, glorot_uniform)
    self.has_bias = use_bias
    if use_bias:
      store.add(name + '/bias', (cout,), zeros_init)
    self.spectral = kind in ('spectral', 'partial_spectral')
    if self.spectral:
      store.add(name + '/u', (1, cout), truncated_normal_init, trainable=False)
    self._copies = {}   # dtype -> (version, wt, wn)
    self.sn = None      # spectral scratch: dict(v, uhat, sig, part, vpart)

  @property
  def kernel(self):
    return self.store[self.name + '/kernel']

  @property
  def bias(self):
    return self.store[self.name + '/bias'] if self.has_bias else None

  def operands(self, ctx: Ctx):
    """Compute-dtype operand copies: wt [cout_assoc][K] and wn [K][cout_assoc]."""
    ent = self._copies.get(ctx.dtype)
    if ent is not None and ent[0] == self.store.version:
      return ent[1], ent[2]
    return self.prep(ctx.dtype, self.store.version)

  def prep(self, dtype, version):
    """Writes the operand copies from the current master weights and stamps them `version` (the
    trainer refreshes a module's copies right behind its Adam update, on the optimiser's side
    stream, stamped with the version the store will have when the step ends)."""
    ent = self._copies.get(dtype)
    w = self.kernel
    K = w.shape[0] * w.shape[1] * w.shape[2]
    co = w.shape[3]
    if ent is None:
      wt = torch.empty((co, K), dtype=dtype, device=w.device)
      wn = torch.empty((K, co), dtype=dtype, device=w.device)
    else:
      wt, wn = ent[1], ent[2]
    _chk(_L().se3ds_weight_prep(w.data_ptr(), K, co, _lib.dtype_code(wt), wt.data_ptr(),
                                wn.data_ptr(), _lib.stream()), 'se3ds_weight_prep')
    self._copies[dtype] = (version, wt, wn)
    return wt, wn


class OperandGroup:
  """The conv layers of one module: their bf16 operand copies refreshed in ONE launch
  (`se3ds_weight_prep_multi`) instead of one launch per layer; layers the batched kernel does not
  take (fp32 compute, thin heads with cout % 4 != 0) keep their own launch."""

  def __init__(self, layers: List['ConvLayer'], dtype, device):
    self.dtype = dtype
    self.batched, self.single, rows, tile0 = [], [], [], 0
    for l in layers:
      w = l.kernel
      K, co = w.shape[0] * w.shape[1] * w.shape[2], w.shape[3]
      if dtype == torch.bfloat16 and K % 8 == 0 and co % 4 == 0:
        ent = l._copies.get(dtype)
        if ent is None:
          wt = torch.empty((co, K), dtype=dtype, device=w.device)
          wn = torch.empty((K, co), dtype=dtype, device=w.device)
          l._copies[dtype] = (-1, wt, wn)
        else:
          wt, wn = ent[1], ent[2]
        if (w.data_ptr() | wt.data_ptr()) % 16 == 0 and wn.data_ptr() % 8 == 0:
          rows.append([w.data_ptr(), K, co, wt.data_ptr(), wn.data_ptr(), tile0])
          tile0 += ((K + 63) // 64) * ((co + 63) // 64)
          self.batched.append(l)
          continue
      self.single.append(l)
    self.tiles = tile0
    self.table = torch.tensor(rows, dtype=torch.int64, device=device) if rows else None

  def prep(self, version):
    if self.table is not None:
      _chk(_L().se3ds_weight_prep_multi(self.table.data_ptr(), len(self.batched), self.tiles,
                                        _lib.stream()), 'se3ds_weight_prep_multi')
      for l in self.batched:
        ent = l._copies[self.dtype]
        l._copies[self.dtype] = (version, ent[1], ent[2])
    for l in self.single:
      l.prep(self.dtype, version)


class SpectralGroup:
  """All spectrally-normalised layers of one model: batched power iteration (layers.py:312-331)
  and the batched gradient fix-up through sigma."""

  def __init__(self, layers: List[ConvLayer], device):
    self.layers = [l for l in layers if l.spectral]
    L = _L()
    nf, pr, vl = (L.se3ds_spectral_table_fields(), L.se3ds_spectral_part_rows(),
                  L.se3ds_spectral_vpart_len())
    rows = []
    for l in self.layers:
      w = l.kernel
      K, co = w.shape[0] * w.shape[1] * w.shape[2], w.shape[3]
      l.sn = dict(v=torch.zeros(K, device=device), uhat=torch.zeros(co, device=device),
                  sig=torch.ones(2, device=device), part=torch.zeros(pr * co, device=device),
                  vpart=torch.zeros(vl, device=device))
      rows.append([w.data_ptr(), l.store[l.name + '/u'].data_ptr(), l.sn['v'].data_ptr(),
                   l.sn['uhat'].data_ptr(), l.sn['sig'].data_ptr(), l.sn['part'].data_ptr(),
                   l.sn['vpart'].data_ptr(), K, co, 0])
      assert len(rows[-1]) == nf
    self._rows = rows
    self.table = torch.tensor(rows, dtype=torch.int64, device=device) if rows else None
    self._eff = [i for i, l in enumerate(self.layers) if l.kind == 'spectral']
    self._eff_table = None

  @property
  def eff_table(self):
    """Rows of the effective SpectralConv layers with their gradient pointers (built on first
    use: touching the gradient views allocates the gradient arena)."""
    if self._eff_table is None and self._eff:
      rows = []
      for i in self._eff:
        l = self.layers[i]
        r = list(self._rows[i])
        r[9] = l.store.grad_views[l.name + '/kernel'].data_ptr()
        rows.append(r)
      self._eff_table = torch.tensor(rows, dtype=torch.int64, device=self.table.device)
    return self._eff_table

  def power_iteration(self, training):
    if self.table is not None:
      _chk(_L().se3ds_spectral_power_iter(self.table.data_ptr(), self.table.shape[0],
                                          1 if training else 0, _lib.stream()),
           'se3ds_spectral_power_iter')

  def tensor_sn(self, store):
    """int64 [number of trainable tensors]: address of the effective spectral layer's table row
    for its kernel tensor, 0 elsewhere (se3ds_multi_sqnorm_sn / se3ds_multi_clip_by_norm_sn)."""
    tab = self.eff_table
    if tab is None:
      return None
    if getattr(self, '_tensor_sn', None) is None:
      nf = tab.shape[1]
      index = {n_: t for t, n_ in enumerate(store.trainable_names)}
      rows = [0] * len(store.trainable_names)
      for j, i in enumerate(self._eff):
        rows[index[self.layers[i].name + '/kernel']] = tab.data_ptr() + 8 * nf * j
      self._tensor_sn = torch.tensor(rows, dtype=torch.int64, device=tab.device)
    return self._tensor_sn

  def backward_fixup(self, prefix=None, dots_only=False):
    """Only layers that convolve with W/sigma (SpectralConv); PartialSpectralConv computes
    sigma but convolves with the raw kernel (layers.py:189-195).  `prefix` restricts the
    fix-up to one top-level module (per-segment gradient synchronisation).  dots_only: just the
    reductions; the fix-up itself then happens inside the clip pass (AdamState.clip_*(fused_sn))."""
    tab = self.eff_table
    if tab is None:
      return
    if prefix is not None:
      cache = self.__dict__.setdefault('_seg_tables', {})
      if prefix not in cache:
        pres = prefix if isinstance(prefix, (tuple, list)) else (prefix,)
        rows = [j for j, i in enumerate(self._eff)
                if any(self.layers[i].name.startswith(pre + '/') for pre in pres)]
        cache[prefix] = tab[rows].contiguous() if rows else None
      tab = cache[prefix]
      if tab is None:
        return
    if dots_only:
      _chk(_L().se3ds_spectral_bwd_dots(tab.data_ptr(), tab.shape[0], _lib.stream()),
           'se3ds_spectral_bwd_dots')
      return
    _chk(_L().se3ds_spectral_bwd_fixup(tab.data_ptr(), tab.shape[0], _lib.stream()),
         'se3ds_spectral_bwd_fixup')


# SE3DS_CHECK_MASKS=1 (debugging): every partial-conv mask that takes a binary-mask fast path
# is checked for {0, 1} values on the device and the host waits for the verdict -- the fast paths
# (zero-page gathers, the shared mask windows, 1 x 1 convs without x * mask) rely on it unchecked.
_CHECK_MASKS = os.environ.get('SE3DS_CHECK_MASKS') == '1'


def mask_window(ctx, mask, n, h, w, ho, wo, k, stride, pad_t, pad_l, wrap, want_bwd):
  if _CHECK_MASKS and ctx.binary_masks and not bool(((mask == 0) | (mask == 1)).all()):
    raise _lib.Se3dsHipError('a partial-conv mask with values outside {0, 1} reached a binary-mask fast path: '
                             'set gan.binary_masks = False (ctx.binary_masks) for fractional masks')
  ratio = torch.empty((n, ho, wo), dtype=torch.float32, device=ctx.device)
  um = torch.empty_like(ratio)
  ru = torch.empty_like(ratio) if want_bwd else None
  bu = torch.empty_like(ratio) if want_bwd else None
  _chk(_L().se3ds_mask_window(mask.data_ptr(), n, h, w, ho, wo, k, k, stride, pad_t, pad_l,
                              1 if wrap else 0, ratio.data_ptr(), um.data_ptr(), _lib.ptr(ru),
                              _lib.ptr(bu), _lib.stream()), 'se3ds_mask_window')
  return ratio, um, ru, bu


def _colsum(ctx, t2d_ptr, dtype_code, rows, c, row_scale=None, groups=1, out=None):
  """Column sums [groups][c] of a [groups][rows][c] tensor (bias / affine gradients); `out`
  (c floats, e.g. a view of the gradient arena) also receives the sums of group 0."""
  sums = torch.empty((groups, 2, c), dtype=torch.float32, device=ctx.device)
  L = _L()
  ws = ctx.ws('norm', L.se3ds_norm_workspace_bytes(groups, c))
  _chk(L.se3ds_norm_stats(t2d_ptr, dtype_code, groups, rows, c, _lib.ptr(row_scale),
                          sums.data_ptr(), _lib.ptr(out), ws.data_ptr(), ws.numel(),
                          _lib.stream()), 'se3ds_norm_stats')
  return sums


_RED_TABLES = {}


def _wgrad(ctx, layer, args, wsz):
  """se3ds_conv2d_wgrad(*args, out_scale=None, accumulate=0, workspace...) -- or, with
  ctx.wgrad_defer, se3ds_conv2d_wgrad_partial into the layer's OWN slab scratch (it must outlive the
  deferred reduction) with the reduction's row appended to the list."""
  L = _L()
  if ctx.wgrad_defer is None:
    ws = _global_ws(ctx.device, 'wgrad', wsz)
    _chk(L.se3ds_conv2d_wgrad(*args, None, 0, ws.data_ptr(), ws.numel(), _lib.stream()),
         'se3ds_conv2d_wgrad')
    return
  ws = layer.__dict__.get('_wgrad_slab')
  if ws is None or ws.numel() < wsz:
    ws = torch.empty((int(wsz),), dtype=torch.uint8, device=ctx.device)
    layer._wgrad_slab = ws
    _RED_TABLES.clear()   # (tables are keyed by slab pointers: the old ones can never match again)
  row = (_lib.c_i64 * 5)()
  _chk(L.se3ds_conv2d_wgrad_partial(*args, ws.data_ptr(), ws.numel(), row, _lib.stream()),
       'se3ds_conv2d_wgrad_partial')
  if row[4]:
    ctx.wgrad_defer.append((row[0], row[1], row[2], row[3]))


def flush_wgrad_reduces(ctx):
  """Runs the deferred split reductions collected so far as one launch on the current stream
  (which must be ordered behind the weight-gradient kernels that wrote the slabs)."""
  rows = ctx.wgrad_defer
  if not rows:
    return
  ctx.wgrad_defer = []   # (a fresh list: `rows` is this launch's, also if a caller still holds it)
  _reduce_rows(ctx, rows)


def reduce_or_defer(ctx, row):
  """row = (slabs, splits, n / 4, destination): dst[i] = sum_s slabs[s][i] -- joins the module's
  deferred reductions (ctx.wgrad_defer) or runs right away as a one-row launch."""
  if ctx.wgrad_defer is not None:
    ctx.wgrad_defer.append(row)
  else:
    _reduce_rows(ctx, [row])


def _reduce_rows(ctx, rows):
  key = (str(ctx.device), tuple(rows))
  ent = _RED_TABLES.get(key)
  if ent is None:   # (pointers are stable: per-layer slabs, one gradient arena -- built once)
    tile = int(_L().se3ds_wgrad_reduce_tile())
    tab, first = [], 0
    for part, splits, n4, dst in rows:
      tab.append([part, splits, n4, dst, first])
      first += (n4 + tile - 1) // tile
    ent = (torch.tensor(tab, dtype=torch.int64, device=ctx.device), first)
    if len(_RED_TABLES) >= 256:   # (varying shapes / batch sizes: bounded, rebuilt on demand)
      _RED_TABLES.clear()
    _RED_TABLES[key] = ent
  # (the bench's instrumented step charges the batched reduce to the weight gradients, as it
  # charges every immediate reduce inside the timed se3ds_conv2d_wgrad call)
  with _Timed('wgrad', 0.0, 'split reduce (batched)'):
    _chk(_L().se3ds_wgrad_reduce_multi(ent[0].data_ptr(), len(rows), ent[1], _lib.stream()),
         'se3ds_wgrad_reduce_multi')


def conv2d(ctx: Ctx, x: Var, layer: ConvLayer, pad=0, wrap=False, mask=None, act=ACT_NONE,
           alpha=0.0):
  """PadLayer(pad, circular=wrap) + conv (+ partial-conv renormalisation / spectral scale /
  bias / activation).  For partial kinds returns (Var, update_mask) (layers.py:140-209)."""
  assert not layer.transpose
  xd = x.data
  n, h, w, cin = xd.shape
  assert cin == layer.cin, (cin, layer.cin)
  k, s = layer.k, layer.stride
  ho, pt = conv_out_size(h, k, s, layer.padding, pad)
  wo, pl = conv_out_size(w, k, s, layer.padding, pad)
  wrap = bool(wrap and pad > 0)
  L = _L()
  wt, wn = layer.operands(ctx)
  partial = layer.kind in ('partial', 'partial_spectral')
  recording = ctx.tape is not None
  ratio = um = ru = bu = None
  in_mask = None
  if partial:
    m = mask
    if m is None:
      m = torch.empty((n, h, w), dtype=torch.float32, device=ctx.device)
      _chk(L.se3ds_fill(m.data_ptr(), _lib.F32, m.numel(), 1.0, _lib.stream()), 'se3ds_fill')
    else:
      in_mask = m
    # The window sums depend on the mask and the geometry only: layers that see the SAME mask tensor
    # with the same geometry share one launch (round 5).  In a ResNet stack the 1x1 convs pass a
    # binary mask through unchanged (update_mask = clip(mask, 0, 1) = mask), so with the alias below
    # conv3 of a block, the next block's conv1 and its downsample conv hit the same entry: 115 ->
    # ~50 launches of 5 us (+ a dependent-launch boundary each) on the encoder's serial path.
    cache = ctx.__dict__.setdefault('_mask_cache', {}) if mask is not None else None
    key = (m.data_ptr(), tuple(m.shape), k, s, pt, pl, wrap, ho, wo, bool(recording))
    ent = cache.get(key) if cache is not None and _MASK_CACHE else None
    if ent is None:
      ratio, um, ru, bu = mask_window(ctx, m, n, h, w, ho, wo, k, s, pt, pl, wrap, recording)
      if (cache is not None and _MASK_CACHE and ctx.binary_masks and k == 1 and s == 1 and pt == 0 and
          pl == 0):
        um = m   # (exactly the kernel's output for a {0, 1} mask: the chain keeps ONE tensor)
      if cache is not None:
        cache[key] = (ratio, um, ru, bu, m)   # (m: keeps the keyed storage alive)
    else:
      ratio, um, ru, bu = ent[:4]
    if in_mask is not None and k == 1 and ctx.binary_masks and _DROP_1X1_MASK:
      # A 1 x 1 partial conv with a {0, 1} mask does not need x * mask: its window IS the pixel, so
      # ratio = update_mask = 0 exactly where the mask is 0 and the epilogue's `* ratio` (and
      # `* update_mask` behind the bias) zeroes those outputs whatever the product was -- PROVIDED the
      # activation there is finite: 0 * Inf and 0 * NaN are NaN, so a non-finite value at a masked pixel,
      # which the input mask used to zero, would now reach y, the fused statistics and (dy = 0, x read
      # unmasked) the weight gradient.  Every input of a 1 x 1 partial conv in this network is a
      # batch-norm + ReLU output or the finite network input, and a non-finite activation anywhere else
      # already poisons the batch statistics of its layer; SE3DS_DROP_1X1_MASK=0 keeps the product.
      # Backward, dy * ratio * update_mask is 0 there, so neither the weight gradient
      # nor dx sees the masked pixels.  Without the mask the kernels skip one dependent global load
      # (mask -> source address of the first LDS-DMA) at the head of every workgroup: ~10 us of a
      # 60 us launch on the encoder's 512 <-> 2048 bottleneck convs.
      in_mask = None
  # SpectralConv convolves with W/(sigma+eps); PartialSpectralConv with the raw kernel.
  scale = layer.sn['sig'][1:] if layer.kind == 'spectral' else None
  bias = layer.bias
  y = ctx.empty((n, ho, wo, layer.cout))
  flops = 2.0 * n * ho * wo * cin * layer.cout * k * k
  tag = f'{k}x{k}s{s} {cin}->{layer.cout} @{ho}x{wo} n{n} {layer.kind}'
  # Training forward: convs that can also emit the column sums of their output (the statistics
  # a following SyncBatchNormalization needs) do so; norm_act picks them up from the Var.
  stats_rows = 0
  # (the sums are taken over the stored, activated outputs, so a fused activation is fine)
  if ctx.training and recording and not getattr(ctx, 'bn_use_moving', False):
    stats_rows = int(L.se3ds_conv2d_fwd_stats_rows(ctx.code, n, cin, ho, wo, layer.cout, k, k, s,
                                                   1 if in_mask is not None else 0,
                                                   1 if ctx.binary_masks else 0))
  stats = None
  with _Timed('fwd', flops, tag):
    if stats_rows > 0:
      stats = torch.empty((stats_rows, 2, layer.cout), dtype=torch.float32, device=ctx.device)
      _chk(L.se3ds_conv2d_fwd_stats(xd.data_ptr(), wt.data_ptr(), y.data_ptr(), ctx.code, n, h, w,
                                    cin, ho, wo, layer.cout, k, k, s, pt, pl, 1 if wrap else 0,
                                    _lib.ptr(in_mask), 1 if ctx.binary_masks else 0,
                                    _lib.ptr(scale), _lib.ptr(bias), _lib.ptr(ratio),
                                    _lib.ptr(um if (partial and bias is not None) else None), act,
                                    float(alpha), stats.data_ptr(), _lib.stream()),
           'se3ds_conv2d_fwd_stats')
    else:
      _chk(L.se3ds_conv2d_fwd(xd.data_ptr(), wt.data_ptr(), y.data_ptr(), ctx.code, n, h, w, cin,
                              ho, wo, layer.cout, k, k, s, pt, pl, 1 if wrap else 0,
                              _lib.ptr(in_mask), 1 if ctx.binary_masks else 0, _lib.ptr(scale),
                              _lib.ptr(bias), _lib.ptr(ratio),
                              _lib.ptr(um if (partial and bias is not None) else None), act,
                              float(alpha), _lib.stream()), 'se3ds_conv2d_fwd')
  out = Var(y)
  if act != ACT_NONE and ctx.act_taps is not None:
    ctx.act_taps[layer.name] = y
  if stats is not None:
    out.col_stats = stats   # [rows][2][cout] partial (sum, sum of squares) of y
  if (recording and partial and bias is not None and ctx.binary_masks and act == ACT_NONE and
      layer.cout % 8 == 0 and xd.dtype == torch.bfloat16):
    out.row_sink = (ru, bu, layer.store, layer.name + '/bias')
  if recording:
    def bwd(n=n):
      dy = out.grad
      pre_scaled = out.grad_scaled is not None and out.grad_scaled == out.gver
      if out.grad_scaled is not None and not pre_scaled:
        # a norm stored its dx already multiplied by this conv's row scale and something else
        # wrote to the gradient afterwards: the sum can no longer be un-mixed -- fail loudly
        raise RuntimeError(f'{layer.name}: pre-scaled output gradient was modified by a second '
                           'consumer (set SE3DS_FUSED_ROW_SCALE=0)')
      out.grad_scaled = None
      out.grad = None
      if dy is None:
        return
      # ctx.batch_limit: this backward pass only concerns the first `limit` samples (their
      # gradients do not depend on the others: no batch statistics on this path)
      lim = getattr(ctx, 'batch_limit', None)
      if lim is not None and lim < n:
        assert not partial and dy.shape[0] == lim and not ctx.param_grads
        n = lim
      if act != ACT_NONE and not out.grad_pre_act:
        _chk(L.se3ds_act_bwd(dy.data_ptr(), y.data_ptr(), ctx.code, dy.numel(), act, float(alpha),
                             dy.data_ptr(), _lib.stream()), 'se3ds_act_bwd')
      out.grad_pre_act = False
      rows = n * ho * wo
      row_scale = None
      dys = dy
      st = layer.store
      bias_done = False
      if partial and pre_scaled:
        # the batch norm behind this conv stored its dx already multiplied by ratio * update_mask
        # and wrote the bias gradient (se3ds_norm_bwd_apply_rows): no pass over dy here
        bias_done = True
      elif partial:
        row_scale = ru if bias is not None else ratio
        if ctx.binary_masks:
          # pre-scale dy once so that wgrad / dgrad can take the LDS-DMA kernels
          dys = ctx.empty(dy.shape)
          if (ctx.param_grads and bias is not None and dy.dtype == torch.bfloat16 and
              layer.cout % 8 == 0):
            # ... and take the bias gradient's column sums from the same read of dy
            sums = torch.empty((1, 2, layer.cout), dtype=torch.float32, device=ctx.device)
            ws = ctx.ws('norm', L.se3ds_norm_workspace_bytes(1, layer.cout))
            _chk(L.se3ds_colsum_row_scale(dy.data_ptr(), ctx.code, rows, layer.cout, bu.data_ptr(),
                                          row_scale.data_ptr(), dys.data_ptr(), sums.data_ptr(),
                                          st.grad_views[layer.name + '/bias'].data_ptr(),
                                          ws.data_ptr(), ws.numel(), _lib.stream()),
                 'se3ds_colsum_row_scale')
            bias_done = True
          else:
            _chk(L.se3ds_row_scale(dy.data_ptr(), ctx.code, rows, layer.cout, row_scale.data_ptr(),
                                   dys.data_ptr(), _lib.stream()), 'se3ds_row_scale')
          row_scale = None
      if ctx.param_grads:
        if bias is not None and not bias_done:
          _colsum(ctx, dy.data_ptr(), ctx.code, rows, layer.cout,
                  row_scale=bu if partial else None, out=st.grad_views[layer.name + '/bias'])
        gk = st.grad_views[layer.name + '/kernel']
        thin_out = (layer.cout <= 16 and cin > 16 and s == 1 and ho == h and wo == w and
                    pt == pl and not wrap and not partial)
        with ctx.on_wgrad_stream(dys, row_scale):   # (scratch below is per stream)
          if thin_out:
            # 128->3 / 128->1 output convs: role-swapped weight gradient (x streamed once)
            wsz = L.se3ds_conv2d_wgrad_swapped_workspace_bytes(n, h, w, cin, layer.cout, k)
            ws = _global_ws(ctx.device, 'wgrad', wsz)
            with _Timed('wgrad', flops, tag):
              _chk(L.se3ds_conv2d_wgrad_swapped(xd.data_ptr(), dys.data_ptr(), gk.data_ptr(),
                                                ctx.code, n, h, w, cin, layer.cout, k, pt, 0,
                                                ws.data_ptr(), ws.numel(), _lib.stream()),
                   'se3ds_conv2d_wgrad_swapped')
          else:
            wsz = L.se3ds_conv2d_wgrad_workspace_bytes(n, ho, wo, cin, layer.cout, k, k)
            with _Timed('wgrad', flops, tag):
              _wgrad(ctx, layer, (xd.data_ptr(), dys.data_ptr(), gk.data_ptr(), ctx.code, n, h,
                                  w, cin, ho, wo, layer.cout, k, k, s, pt, pl,
                                  1 if wrap else 0, _lib.ptr(in_mask),
                                  1 if ctx.binary_masks else 0, _lib.ptr(row_scale)), wsz)
      if x.requires_grad:
        prev = x.grad
        shape = (n,) + tuple(xd.shape[1:])
        have_prev = prev is not None and tuple(prev.shape) == shape and prev.dtype == xd.dtype
        bn_rows = 0
        if (x.bn_src is not None and row_scale is None and lim is None and _FUSED_BN_BWD and
            (prev is None or have_prev)):
          bn_rows = int(L.se3ds_conv2d_dgrad_bnstats_rows(ctx.code, n, h, w, cin, layer.cout, k, k,
                                                          s, 0))
        if bn_rows > 0:
          # x is a batch norm's output: this data gradient also takes the norm's backward
          # statistics from the gradient it stores (speculatively: they are used only if no later
          # contribution changes x.grad -- the first consumer in forward order is the last here)
          bx, bmask, bmean, brstd, bact, balpha = x.bn_src
          stats = torch.empty((bn_rows, 2, cin), dtype=torch.float32, device=ctx.device)
          if have_prev:
            _grad_wait(x)
            dx = prev
          else:
            dx = ctx.empty(shape)
          with _Timed('dgrad', flops, tag):
            _chk(L.se3ds_conv2d_dgrad_bnstats(
                dys.data_ptr(), wn.data_ptr(), dx.data_ptr(), ctx.code, n, h, w, cin, ho, wo,
                layer.cout, k, k, s, pt, pl, 1 if wrap else 0, _lib.ptr(scale), _lib.ptr(in_mask),
                prev.data_ptr() if have_prev else None, bx.data_ptr(), _lib.ptr(bmask),
                bmean.data_ptr(), brstd.data_ptr(), bact, float(balpha), stats.data_ptr(),
                _lib.stream()), 'se3ds_conv2d_dgrad_bnstats')
          if have_prev:
            x.gver += 1
            _grad_mark(x)
          else:
            accumulate(x, dx)
          x.bn_stats = (stats, x.gver)
        elif have_prev:
          # second contribution (e.g. a ResNet block's input: residual branch first, then this
          # conv): the epilogue adds the existing gradient in place instead of a separate pass
          _grad_wait(x)
          with _Timed('dgrad', flops * n / xd.shape[0], tag):
            _chk(L.se3ds_conv2d_dgrad_acc(dys.data_ptr(), wn.data_ptr(), prev.data_ptr(), ctx.code,
                                          n, h, w, cin, ho, wo, layer.cout, k, k, s, pt, pl,
                                          1 if wrap else 0, _lib.ptr(row_scale), _lib.ptr(scale),
                                          None, _lib.ptr(in_mask), ACT_NONE, 0.0, prev.data_ptr(),
                                          _lib.stream()), 'se3ds_conv2d_dgrad_acc')
          x.gver += 1
          _grad_mark(x)
        else:
          dx = ctx.empty(shape)
          with _Timed('dgrad', flops * n / xd.shape[0], tag):
            _chk(L.se3ds_conv2d_dgrad(dys.data_ptr(), wn.data_ptr(), dx.data_ptr(), ctx.code, n, h,
                                      w, cin, ho, wo, layer.cout, k, k, s, pt, pl,
                                      1 if wrap else 0, _lib.ptr(row_scale), _lib.ptr(scale), None,
                                      _lib.ptr(in_mask), ACT_NONE, 0.0, _lib.stream()),
                 'se3ds_conv2d_dgrad')
          accumulate(x, dx)
    ctx.record(bwd)
  if partial:
    return out, um
  return out


def conv_transpose2d(ctx: Ctx, x: Var, layer: ConvLayer):
  """Keras Conv2DTranspose, stride 2: k3 'SAME' output_padding=1, k2 'VALID', k2 'SAME'
  (layers.py:417-423,475-480; image_models.py:440-441); all produce 2H x 2W."""
  assert layer.transpose and layer.stride == 2 and layer.k in (2, 3)
  xd = x.data
  n, hi, wi, cin_t = xd.shape
  assert cin_t == layer.cin
  k = layer.k
  H, W = 2 * hi, 2 * wi          # the associated forward conv maps (H,W,cout_T) -> (hi,wi,cin_T)
  L = _L()
  wt, wn = layer.operands(ctx)   # kernel (k,k,cout_T,cin_T) == HWIO of the associated conv
  y = ctx.empty((n, H, W, layer.cout))
  bias = layer.bias
  flops = 2.0 * n * hi * wi * cin_t * layer.cout * k * k
  tag = f'convT {k}x{k}s2 {cin_t}->{layer.cout} @{hi}x{wi}->{H}x{W} n{n}'
  with _Timed('convT_fwd', flops, tag):
    if k == 2 and layer.cout % 4 == 0 and _CONVT_2X2:
      # every output pixel sees one tap: two 1x1 convolutions with (kx, co) as channels, 512-byte
      # output runs (se3ds_conv_transpose2x2_fwd) instead of four parity-class passes
      _chk(L.se3ds_conv_transpose2x2_fwd(xd.data_ptr(), wn.data_ptr(), y.data_ptr(), ctx.code, n, hi,
                                         wi, cin_t, layer.cout, _lib.ptr(bias), _lib.stream()),
           'se3ds_conv_transpose2x2_fwd')
    else:
      _chk(L.se3ds_conv2d_dgrad(xd.data_ptr(), wn.data_ptr(), y.data_ptr(), ctx.code, n, H, W,
                                layer.cout, hi, wi, cin_t, k, k, 2, 0, 0, 0, None, None,
                                _lib.ptr(bias), None, ACT_NONE, 0.0, _lib.stream()),
           'se3ds_conv2d_dgrad')
  out = Var(y)
  if ctx.tape is not None:
    def bwd():
      dy = out.grad
      out.grad = None
      if dy is None:
        return
      st = layer.store
      if ctx.param_grads:
        if bias is not None:
          _colsum(ctx, dy.data_ptr(), ctx.code, n * H * W, layer.cout,
                  out=st.grad_views[layer.name + '/bias'])
        wsz = L.se3ds_conv2d_wgrad_workspace_bytes(n, hi, wi, layer.cout, cin_t, k, k)
        gk = st.grad_views[layer.name + '/kernel']
        with _Timed('convT_wgrad', flops, tag):
          _wgrad(ctx, layer, (dy.data_ptr(), xd.data_ptr(), gk.data_ptr(), ctx.code, n, H, W,
                              layer.cout, hi, wi, cin_t, k, k, 2, 0, 0, 0, None, 0, None), wsz)
      if x.requires_grad:
        dx = ctx.empty(xd.shape)
        with _Timed('convT_dgrad', flops, tag):
          _chk(L.se3ds_conv2d_fwd(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), ctx.code, n, H, W,
                                  layer.cout, hi, wi, cin_t, k, k, 2, 0, 0, 0, None, 0, None, None,
                                  None, None, ACT_NONE, 0.0, _lib.stream()), 'se3ds_conv2d_fwd')
        accumulate(x, dx)
    ctx.record(bwd)
  return out


# ------------------------------------------------------------------------------------ norms
class NormLayer:
  """SyncBatchNormalization (kind='batch') or tfa InstanceNormalization (kind='instance')."""

  def __init__(self, store: ParamStore, name, c, kind='batch'):
    self.store, self.name, self.c, self.kind = store, name, c, kind
    store.add(name + '/gamma', (c,), ones_init)
    store.add(name + '/beta', (c,), zeros_init)
    if kind == 'batch':
      store.add(name + '/moving_mean', (c,), zeros_init, trainable=False)
      store.add(name + '/moving_variance', (c,), ones_init, trainable=False)


_NORM_DEBUG = {} if os.environ.get('SE3DS_NORM_DEBUG') else None
# SE3DS_FUSED_BN_BWD=1: batch-norm backward statistics from the epilogue of the data gradient that
# produces dy (se3ds_conv2d_dgrad_bnstats) instead of their own pass over dy and x.  OFF by
# default: measured 2.8-4.4 ms per step SLOWER at batch 8 (the x tile is then read at the tail of
# every workgroup's epilogue -- data gradient 887 -> 768 TFLOP/s, +8.5 ms -- to save a streaming
# pass that costs 5.7 ms).  Kept, tested, for larger batches / tensors.
_FUSED_BN_BWD = os.environ.get('SE3DS_FUSED_BN_BWD', '0') == '1'
# SE3DS_FUSED_ROW_SCALE=0: the backward pass of a biased partial conv always takes dy * (ratio *
# update_mask) and its bias gradient from a pass of its own over dy (se3ds_colsum_row_scale);
# default: the batch norm in front of it (in backward order) stores its dx pre-scaled and writes
# the bias gradient from the same kernel (se3ds_norm_bwd_apply_rows)
_FUSED_ROW_SCALE = os.environ.get('SE3DS_FUSED_ROW_SCALE', '1') != '0'
# SE3DS_NORM_CG=0: batch-norm backward as statistics + column reduction + apply (three launches);
# default: se3ds_norm_bwd_cg (two launches, the apply workgroups fold the partial rows) where the
# shape allows (bf16, channels % 64 == 0, >= 512 channels, one replica)
_NORM_CG = os.environ.get('SE3DS_NORM_CG', '1') != '0'
# SE3DS_CONVT_2X2=0: 2x2 stride-2 transposed convs through the parity-class data-gradient kernel
_CONVT_2X2 = os.environ.get('SE3DS_CONVT_2X2', '1') != '0'
# SE3DS_MASK_CACHE=0: every partial conv launches its own mask-window kernel
_MASK_CACHE = os.environ.get('SE3DS_MASK_CACHE', '1') != '0'
# SE3DS_DROP_1X1_MASK=0: 1 x 1 partial convs with a binary mask still multiply their input by it
_DROP_1X1_MASK = os.environ.get('SE3DS_DROP_1X1_MASK', '1') != '0'


def norm_act(ctx: Ctx, x: Var, layer: NormLayer, act=ACT_NONE, alpha=0.0, res: Var = None,
             post: Var = None, in_act=None) -> Var:
  """y = act(norm(x) [+ res]) [+ post].  Batch norm uses cross-replica batch statistics when
  training (SyncBatchNormalization) and moving statistics otherwise.  in_act = (act, alpha):
  x is the output of a conv with that fused activation and this norm is its ONLY consumer; the
  backward then applies the activation derivative itself (it reads x anyway) and the conv skips
  its separate derivative pass."""
  xd = x.data
  n, h, w, c = xd.shape
  st = layer.store
  L = _L()
  inst = layer.kind == 'instance'
  g = n if inst else 1
  r = h * w if inst else n * h * w
  eps = IN_EPS if inst else BN_EPS
  gamma, beta = st[layer.name + '/gamma'], st[layer.name + '/beta']
  use_moving = (not inst) and (not ctx.training or getattr(ctx, 'bn_use_moving', False))
  scale = torch.empty((g, c), dtype=torch.float32, device=ctx.device)
  shift = torch.empty_like(scale)
  mean = torch.empty_like(scale)
  rstd = torch.empty_like(scale)
  count = float(r)
  if use_moving:
    _chk(L.se3ds_norm_finalize(None, 1.0, g, c, gamma.data_ptr(), beta.data_ptr(), eps,
                               BN_MOMENTUM, st[layer.name + '/moving_mean'].data_ptr(),
                               st[layer.name + '/moving_variance'].data_ptr(), 1,
                               scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                               rstd.data_ptr(), _lib.stream()), 'se3ds_norm_finalize')
  else:
    fused = getattr(x, 'col_stats', None)
    finalized = False
    if (fused is not None and not inst and fused.shape[2] == c and ctx.world == 1 and
        fused.shape[0] <= 2048):
      # statistics came out of the producing convolution's epilogue; single replica: column
      # reduction and finalize in one launch
      _chk(L.se3ds_norm_reduce_rows_finalize(
          fused.data_ptr(), fused.shape[0], c, count, gamma.data_ptr(), beta.data_ptr(), eps,
          BN_MOMENTUM, st[layer.name + '/moving_mean'].data_ptr(),
          st[layer.name + '/moving_variance'].data_ptr(), scale.data_ptr(), shift.data_ptr(),
          mean.data_ptr(), rstd.data_ptr(), _lib.stream()), 'se3ds_norm_reduce_rows_finalize')
      x.col_stats = None
      finalized = True
    elif fused is not None and not inst and fused.shape[2] == c:
      # statistics came out of the producing convolution's epilogue
      sums = torch.empty((1, 2, c), dtype=torch.float32, device=ctx.device)
      ws = ctx.ws('norm', L.se3ds_norm_workspace_bytes(max(1, (fused.shape[0] + 511) // 512), c))
      _chk(L.se3ds_norm_reduce_rows(fused.data_ptr(), fused.shape[0], c, sums.data_ptr(),
                                    ws.data_ptr(), ws.numel(), _lib.stream()),
           'se3ds_norm_reduce_rows')
      x.col_stats = None
    else:
      if _NORM_DEBUG is not None:   # SE3DS_NORM_DEBUG: which norms take their own statistics pass
        key = (layer.name, tuple(xd.shape), layer.kind)
        _NORM_DEBUG[key] = _NORM_DEBUG.get(key, 0) + 1
      sums = _colsum(ctx, xd.data_ptr(), ctx.code, r, c, groups=g)
    if not finalized:
      if not inst and ctx.world > 1:
        ctx.allreduce_sum(sums)
        count = float(r * ctx.world)
      mm = st[layer.name + '/moving_mean'] if not inst else None
      mv = st[layer.name + '/moving_variance'] if not inst else None
      _chk(L.se3ds_norm_finalize(sums.data_ptr(), count, g, c, gamma.data_ptr(), beta.data_ptr(),
                                 eps, BN_MOMENTUM, _lib.ptr(mm), _lib.ptr(mv), 0, scale.data_ptr(),
                                 shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _lib.stream()),
           'se3ds_norm_finalize')
  y = ctx.empty(xd.shape)
  # one "output > 0" bit per element for the backward pass (instead of re-reading y twice)
  amask = None
  if (ctx.tape is not None and act != ACT_NONE and post is None and c % 8 == 0 and
      xd.dtype == torch.bfloat16):
    amask = torch.empty(xd.numel() // 8, dtype=torch.uint8, device=ctx.device)
  _chk(L.se3ds_norm_apply(xd.data_ptr(), ctx.code, g, r, c, scale.data_ptr(), shift.data_ptr(),
                          _lib.ptr(res.data if res is not None else None),
                          _lib.ptr(post.data if post is not None else None), act, float(alpha),
                          y.data_ptr(), _lib.ptr(amask), _lib.stream()), 'se3ds_norm_apply')
  out = Var(y)
  if act != ACT_NONE and ctx.act_taps is not None:
    ctx.act_taps[layer.name] = y
  if (ctx.tape is not None and not inst and not use_moving and post is None and c % 8 == 0 and
      xd.dtype == torch.bfloat16 and (act == ACT_NONE or amask is not None)):
    out.bn_src = (xd, amask, mean, rstd, act, alpha)   # a consuming conv's data gradient may fuse
  sync_bwd = (not inst) and ctx.world > 1 and not use_moving   # the backward all-reduces too
  if ctx.tape is not None:
    def bwd(g=g):
      dy = out.grad
      fused = out.bn_stats
      if fused is not None and fused[1] != out.gver:
        fused = None   # the gradient changed after the epilogue that took the statistics
      out.bn_stats = None
      out.grad = None
      if dy is None:
        if sync_bwd:
          # no gradient reaches this norm on this replica / branch: its statistics gradients are
          # zero, but the collective (and a paired twin parked in allreduce_then) still needs us
          ctx.allreduce_then(torch.zeros((g, 2, c), dtype=torch.float32, device=ctx.device),
                             lambda: None)
        return
      if post is not None:
        # y = act(.) + post: the activation mask must come from (y - post); recompute it
        raise NotImplementedError
      lim = getattr(ctx, 'batch_limit', None)
      shape = xd.shape
      if lim is not None and lim < n:
        # per-sample statistics only (instance norm): the first `lim` groups are a prefix of
        # every buffer involved
        assert inst and dy.shape[0] == lim and not ctx.param_grads
        g = lim
        shape = (lim,) + tuple(xd.shape[1:])
      ws = ctx.ws('norm', L.se3ds_norm_workspace_bytes(g, c))
      want_res = res is not None and res.requires_grad
      dres = ctx.empty(shape) if want_res else None
      dx = ctx.empty(shape)
      if use_moving:
        if ctx.param_grads:
          bs = torch.empty((g, 2, c), dtype=torch.float32, device=ctx.device)
          _chk(L.se3ds_norm_bwd_stats(dy.data_ptr(), y.data_ptr(), xd.data_ptr(), ctx.code, g, r,
                                      c, mean.data_ptr(), rstd.data_ptr(), act, float(alpha),
                                      bs.data_ptr(), st.grad_views[layer.name + '/beta'].data_ptr(),
                                      st.grad_views[layer.name + '/gamma'].data_ptr(),
                                      _lib.ptr(amask), ws.data_ptr(), ws.numel(), _lib.stream()),
               'se3ds_norm_bwd_stats')
        _chk(L.se3ds_affine_bwd(dy.data_ptr(), y.data_ptr(), ctx.code, g, r, c, scale.data_ptr(),
                                act, float(alpha), dx.data_ptr(), _lib.ptr(dres), _lib.ptr(amask),
                                _lib.stream()), 'se3ds_affine_bwd')
      else:
        in_a = in_act[0] if in_act else 0
        if (_NORM_CG and not sync_bwd and g == 1 and fused is None and lim is None and
            L.se3ds_norm_bwd_cg_supported(ctx.code, r, c, act, 1 if amask is not None else 0, in_a)):
          # Round 5: statistics partials + apply in the channel-group layout, the apply workgroups
          # fold the partial rows themselves -- no stand-alone column reduction between the passes
          # (se3ds_norm_bwd_cg); beta / gamma gradients are written by the apply kernel
          sink = x.row_sink
          use_rows = (sink is not None and _FUSED_ROW_SCALE and ctx.param_grads and not in_act and
                      x.grad is None)
          cws = ctx.ws('norm_cg', L.se3ds_norm_bwd_cg_workspace_bytes(c))
          colpart, nrows = None, 0
          if use_rows:
            # (the bias partials must outlive the module's deferred reduction: the layer's own)
            nrows = int(L.se3ds_norm_bwd_cg_col_rows(r, c))
            colpart = layer.__dict__.get('_cg_colpart')
            if colpart is None or colpart.numel() < nrows * c:
              colpart = torch.empty(nrows * c, dtype=torch.float32, device=ctx.device)
              layer._cg_colpart = colpart
          pg = ctx.param_grads
          _chk(L.se3ds_norm_bwd_cg(
              dy.data_ptr(), xd.data_ptr(), ctx.code, r, c, mean.data_ptr(), rstd.data_ptr(),
              gamma.data_ptr(), count, act, float(alpha), dx.data_ptr(), _lib.ptr(dres),
              _lib.ptr(amask), in_a, float(in_act[1]) if in_act else 0.0,
              st.grad_views[layer.name + '/beta'].data_ptr() if pg else None,
              st.grad_views[layer.name + '/gamma'].data_ptr() if pg else None, None,
              sink[1].data_ptr() if use_rows else None, sink[0].data_ptr() if use_rows else None,
              _lib.ptr(colpart), cws.data_ptr(), cws.numel(), _lib.stream()), 'se3ds_norm_bwd_cg')
          if _NORM_DEBUG is not None:
            for k in (('cg', tuple(xd.shape), layer.kind),) + (
                (('fused-rows', tuple(xd.shape), layer.kind),) if use_rows else ()):
              _NORM_DEBUG[k] = _NORM_DEBUG.get(k, 0) + 1
          if use_rows:
            reduce_or_defer(ctx, (colpart.data_ptr(), nrows, c // 4,
                                  sink[2].grad_views[sink[3]].data_ptr()))
          if in_act:
            x.grad_pre_act = True
          accumulate(x, dx)
          if use_rows:
            x.grad_scaled = x.gver
          if want_res:
            accumulate(res, dres)
          return
        bs = torch.empty((g, 2, c), dtype=torch.float32, device=ctx.device)
        direct = ctx.param_grads and g == 1
        # parameter gradients are the LOCAL sums (aggregated later with every other gradient);
        # for batch norm the reduction writes them straight into the gradient arena
        if fused is not None and g == 1 and fused[0].shape[2] == c:
          # the statistics came out of the data-gradient epilogue that produced dy
          rows = fused[0].shape[0]
          rws = ctx.ws('norm', L.se3ds_norm_workspace_bytes(max(1, (rows + 511) // 512), c))
          _chk(L.se3ds_norm_reduce_rows_dst(
              fused[0].data_ptr(), rows, c, bs.data_ptr(),
              st.grad_views[layer.name + '/beta'].data_ptr() if direct else None,
              st.grad_views[layer.name + '/gamma'].data_ptr() if direct else None,
              rws.data_ptr(), rws.numel(), _lib.stream()), 'se3ds_norm_reduce_rows_dst')
          if _NORM_DEBUG is not None:
            _NORM_DEBUG[('fused-bwd', tuple(xd.shape), layer.kind)] = \
                _NORM_DEBUG.get(('fused-bwd', tuple(xd.shape), layer.kind), 0) + 1
        else:
          _chk(L.se3ds_norm_bwd_stats(
              dy.data_ptr(), y.data_ptr(), xd.data_ptr(), ctx.code, g, r, c, mean.data_ptr(),
              rstd.data_ptr(), act, float(alpha), bs.data_ptr(),
              st.grad_views[layer.name + '/beta'].data_ptr() if direct else None,
              st.grad_views[layer.name + '/gamma'].data_ptr() if direct else None,
              _lib.ptr(amask), ws.data_ptr(), ws.numel(), _lib.stream()), 'se3ds_norm_bwd_stats')
        if not ctx.param_grads or direct:
          pass
        else:
          tot = _colsum(ctx, bs.data_ptr(), _lib.F32, g, 2 * c)
          st.grad_views[layer.name + '/beta'].copy_(tot[0, 0, :c])
          st.grad_views[layer.name + '/gamma'].copy_(tot[0, 0, c:])

        def finish():
          sink = x.row_sink
          rows_done = False
          if (sink is not None and _FUSED_ROW_SCALE and ctx.param_grads and g == 1 and not in_act and
              x.grad is None and lim is None and xd.dtype == torch.bfloat16):
            # x is the output of a biased partial conv and this norm (so far) its only consumer:
            # store dx pre-scaled for that conv's backward pass and write its bias gradient here
            cws = ctx.ws('norm_rows', L.se3ds_norm_workspace_bytes(3, c))
            rc = L.se3ds_norm_bwd_apply_rows(
                dy.data_ptr(), xd.data_ptr(), ctx.code, r, c, mean.data_ptr(), rstd.data_ptr(),
                gamma.data_ptr(), bs.data_ptr(), count, act, float(alpha), dx.data_ptr(),
                _lib.ptr(dres), _lib.ptr(amask), sink[1].data_ptr(), sink[0].data_ptr(),
                sink[2].grad_views[sink[3]].data_ptr(), cws.data_ptr(), cws.numel(), _lib.stream())
            if rc == 0:
              rows_done = True
              if _NORM_DEBUG is not None:
                _NORM_DEBUG[('fused-rows', tuple(xd.shape), layer.kind)] = \
                    _NORM_DEBUG.get(('fused-rows', tuple(xd.shape), layer.kind), 0) + 1
            elif rc not in (-5, -3):   # SE3DS_E_UNSUPPORTED / SE3DS_E_WORKSPACE: separate passes
              _chk(rc, 'se3ds_norm_bwd_apply_rows')
          if not rows_done:
            _chk(L.se3ds_norm_bwd_apply(dy.data_ptr(), y.data_ptr(), xd.data_ptr(), ctx.code, g, r,
                                        c, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                        bs.data_ptr(), count, act, float(alpha), dx.data_ptr(),
                                        _lib.ptr(dres), _lib.ptr(amask),
                                        in_act[0] if in_act else 0,
                                        float(in_act[1]) if in_act else 0.0,
                                        _lib.stream()), 'se3ds_norm_bwd_apply')
          if in_act:
            x.grad_pre_act = True
          accumulate(x, dx)
          if rows_done:
            x.grad_scaled = x.gver
          if want_res:
            accumulate(res, dres)
        if sync_bwd:
          ctx.allreduce_then(bs, finish)   # SyncBN backward: statistics gradients are global
        else:
          finish()
        return
      accumulate(x, dx)
      if want_res:
        accumulate(res, dres)
    ctx.record(bwd, sync=sync_bwd)
  return out


def add(ctx: Ctx, a: Var, b: Var) -> Var:
  """out = a + b (decoder skip connections, image_models.py:455-482)."""
  y = ctx.empty(a.data.shape)
  _chk(_L().se3ds_add(a.data.data_ptr(), b.data.data_ptr(), ctx.code, y.numel(), y.data_ptr(),
                      _lib.stream()), 'se3ds_add')
  out = Var(y)
  if ctx.tape is not None:
    def bwd():
      dy = out.grad
      out.grad = None
      if dy is None:
        return
      if a.requires_grad and b.requires_grad and a.grad is None and b.grad is None:
        # both take the same gradient tensor; give b its own copy so later in-place
        # accumulations into one do not leak into the other
        a.grad = dy
        b.grad = dy.clone()
      else:
        accumulate(a, dy)
        accumulate(b, dy if a.grad is not dy else dy.clone())
    ctx.record(bwd)
  return out


# ------------------------------------------------------------------------- pooling / resize
def maxpool2x2(ctx: Ctx, x: Var) -> Var:
  xd = x.data
  n, h, w, c = xd.shape
  y = ctx.empty((n, (h + 1) // 2, (w + 1) // 2, c))
  L = _L()
  _chk(L.se3ds_maxpool2x2_fwd(xd.data_ptr(), ctx.code, n, h, w, c, y.data_ptr(), _lib.stream()),
       'se3ds_maxpool2x2_fwd')
  out = Var(y)
  if ctx.tape is not None:
    def bwd():
      dy = out.grad
      out.grad = None
      if dy is None or not x.requires_grad:
        return
      dx = ctx.empty(xd.shape)
      _chk(L.se3ds_maxpool2x2_bwd(dy.data_ptr(), xd.data_ptr(), y.data_ptr(), ctx.code, n, h, w, c,
                                  dx.data_ptr(), _lib.stream()), 'se3ds_maxpool2x2_bwd')
      accumulate(x, dx)
    ctx.record(bwd)
  return out


def maxpool2x2_mask(ctx: Ctx, m: torch.Tensor) -> torch.Tensor:
  """MaxPool of the (N,H,W) fp32 mask (image_models.py:289); no gradient."""
  n, h, w = m.shape
  y = torch.empty((n, (h + 1) // 2, (w + 1) // 2), dtype=torch.float32, device=m.device)
  _chk(_L().se3ds_maxpool2x2_fwd(m.data_ptr(), _lib.F32, n, h, w, 1, y.data_ptr(), _lib.stream()),
       'se3ds_maxpool2x2_fwd')
  return y


def avgpool3s2(ctx: Ctx, x: Var) -> Var:
  xd = x.data
  n, h, w, c = xd.shape
  y = ctx.empty((n, (h + 1) // 2, (w + 1) // 2, c))
  L = _L()
  _chk(L.se3ds_avgpool3s2_fwd(xd.data_ptr(), ctx.code, n, h, w, c, y.data_ptr(), _lib.stream()),
       'se3ds_avgpool3s2_fwd')
  out = Var(y, x.requires_grad)
  if ctx.tape is not None:
    def bwd(n=n):
      dy = out.grad
      out.grad = None
      if dy is None or not x.requires_grad:
        return
      lim = getattr(ctx, 'batch_limit', None)
      if lim is not None and lim < n:
        assert dy.shape[0] == lim
        n = lim
      dx = ctx.empty((n,) + tuple(xd.shape[1:]))
      _chk(L.se3ds_avgpool3s2_bwd(dy.data_ptr(), ctx.code, n, h, w, c, dx.data_ptr(),
                                  _lib.stream()), 'se3ds_avgpool3s2_bwd')
      accumulate(x, dx)
    ctx.record(bwd)
  return out


def upsample2x(ctx: Ctx, x: Var) -> Var:
  xd = x.data
  n, h, w, c = xd.shape
  y = ctx.empty((n, 2 * h, 2 * w, c))
  L = _L()
  _chk(L.se3ds_upsample2x_fwd(xd.data_ptr(), ctx.code, n, h, w, c, y.data_ptr(), _lib.stream()),
       'se3ds_upsample2x_fwd')
  out = Var(y)
  if ctx.tape is not None:
    def bwd():
      dy = out.grad
      out.grad = None
      if dy is None or not x.requires_grad:
        return
      dx = ctx.empty(xd.shape)
      _chk(L.se3ds_upsample2x_bwd(dy.data_ptr(), ctx.code, n, h, w, c, dx.data_ptr(),
                                  _lib.stream()), 'se3ds_upsample2x_bwd')
      accumulate(x, dx)
    ctx.record(bwd)
  return out


def pad2d(x: torch.Tensor, pad: int, circular: bool, mode='CONSTANT', constant_value=0.0):
  """Standalone PadLayer (models/layers.py:22-97) on an NHWC fp32/bf16 tensor."""
  _lib.require_cuda(x)
  n, h, w, c = x.shape
  x = x.contiguous()
  y = torch.empty((n, h + 2 * pad, w + 2 * pad, c), dtype=x.dtype, device=x.device)
  code = {'CONSTANT': 0, 'REFLECT': 1, 'SYMMETRIC': 2}[mode.upper()]
  _chk(_L().se3ds_pad2d(x.data_ptr(), _lib.dtype_code(x), n, h, w, c, pad, code,
                        1 if circular else 0, float(constant_value), y.data_ptr(), _lib.stream()),
       'se3ds_pad2d')
  return y


# ---------------------------------------------------------------------------------- heads
def head(ctx: Ctx, x: Var, kind: int):
  """kind 0: rgb = (tanh(x)+1)/2 ; kind 1: depth = clip(x, 0, 1)  (image_models.py:187-190).
  Returns the fp32 output tensor and a function taking its fp32 gradient."""
  xd = x.data
  y = torch.empty(xd.shape, dtype=torch.float32, device=ctx.device)
  L = _L()
  _chk(L.se3ds_head_fwd(xd.data_ptr(), ctx.code, xd.numel(), kind, y.data_ptr(), _lib.stream()),
       'se3ds_head_fwd')
  def push_grad(dy: torch.Tensor):
    dx = ctx.empty(xd.shape)
    _chk(L.se3ds_head_bwd(dy.data_ptr(), y.data_ptr(), xd.data_ptr(), ctx.code, xd.numel(), kind,
                          dx.data_ptr(), _lib.stream()), 'se3ds_head_bwd')
    accumulate(x, dx)
  return y, push_grad
