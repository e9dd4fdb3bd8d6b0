"""Debug: where do the fused clip+Adam pass and the separate passes differ (whole arena)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_nets_gpu import _make_gan, DEV
gan = _make_gan(64, 8, '50', 3)
G, opt = gan.generator, gan.g_optimizer
st = G.store
st.grad_views
G.spectral.power_iteration(training=True)
ema = gan.ema_generator.store.theta
nt = len(st.trainable_names)
gen = torch.Generator(device=DEV).manual_seed(11)
theta0, m0, v0, ema0 = st.theta.clone(), opt.m.clone(), opt.v.clone(), ema.clone()
it0 = opt.iterations
g0 = torch.randn(st.grad.shape, generator=gen, device=DEV) * 10.0
def run(fused, per_segment):
  st.theta.copy_(theta0); opt.m.copy_(m0); opt.v.copy_(v0); ema.copy_(ema0); opt.iterations = it0
  st.grad.copy_(g0)
  opt.begin_step()
  segs = st.segments(G.SEGMENTS)
  todo = list(segs.items()) if per_segment else [('all', (0, nt, 0, st.theta.numel()))]
  for seg, (t0, t1, e0, e1) in todo:
    G.spectral.backward_fixup(prefix=G.SEGMENTS[seg] if per_segment else None, dots_only=True)
    if fused:
      assert opt.clip_apply(t0, t1, 5.0, True, ema, 1e-3)
    else:
      opt.clip_segment(t0, t1, 5.0, fused_sn=True)
      opt.apply_segment(e0, e1, ema, 1e-3)
  opt.end_step()
  torch.cuda.synchronize()
  return st.theta.clone(), opt.m.clone(), opt.v.clone(), opt.sqnorm.clone()
res = {(f, p): run(f, p) for f in (False, True) for p in (True, False)}
ref = res[(False, True)]
for key, val in res.items():
  print('fused=%s per_segment=%s:' % key, 'theta max diff vs (unfused, per-segment) %.3e' % float((val[0] - ref[0]).abs().max()),
        'm %.3e' % float((val[1] - ref[1]).abs().max()), 'sqnorm rel %.3e' % float(((val[3] - ref[3]).abs() / ref[3].clamp_min(1e-30)).max()))
a, b = res[(True, False)], res[(False, False)]
for t, name in enumerate(st.trainable_names):
  o, n, shape = st._off_tr[name]
  d = float((a[0][o:o + n] - b[0][o:o + n]).abs().max())
  if d > 0:
    print('  whole arena fused vs unfused:', name, shape, 'theta diff %.3e' % d, 'm diff %.3e' % float((a[1][o:o + n] - b[1][o:o + n]).abs().max()),
          'sq', float(a[3][t]), float(b[3][t]))
