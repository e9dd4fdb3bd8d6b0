"""BASELINE.json configurations, one test each, on the real dimensions (ResNet-101, gen_dims 128):

  cfg1  configs/lowres/lowres.gin unmodified: one full train_g_d at 128x256, batch 2 vs the oracle
        (clipped gradients of all 1.1 B parameters, Adam update, BN / spectral state, metrics),
        fp32 path and bf16 path on identical weights.
  cfg2  256x512 fp32 inference: SE3DSModel (warp + generator, circular padding, moving statistics)
        vs oracle/model_np.py, incl. autoregressive feedback frames (models.py:247-366).
  cfg3  configs/highres/highres.gin at 512x1024 bf16: loss values of a batch-1 step vs the oracle's
        forward, finiteness, masked-pixel invariance (layers_test.py:64-86) on the HIP path.
  cfg5  1024x2048, 2 source views: unproject + project/splat bit-exact vs the C oracle.
(cfg4 = cfg3 on 8 GPUs: needs hardware the tests do not have.)
"""
import os
import time

import numpy as np
import pytest
import torch

from oracle import model_np
from oracle import nets_torch as O
from oracle import warp_c
from oracle import warp_np
from se3ds_amd import gin_lite
from se3ds_amd.hipops import nn
from se3ds_amd.models import image_models, layers, model_config, models
from se3ds_amd.trainers import gan_manager, se3ds_trainer
from se3ds_amd.utils import pano_utils

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32 = np.float32


def rel_err(a, b):
  a = np.asarray(a, np.float64)
  b = np.asarray(b, np.float64)
  return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def synth_batch(n, h, seed=1234):
  """SURVEY 8d synthetic inputs."""
  g = torch.Generator().manual_seed(seed)
  w = 2 * h
  image = torch.rand((n, h, w, 3), generator=g)
  depth = torch.rand((n, h, w, 1), generator=g)
  poison = torch.rand((n, h, w, 1), generator=g)
  depth = torch.where(poison < 0.02, torch.zeros_like(depth), depth)
  depth = torch.where(poison > 0.99, torch.ones_like(depth), depth)
  pm = (torch.rand((n, h, w, 1), generator=g) < 0.5).float()
  pm[:, h // 3:h // 3 + max(1, h // 8)] = 0
  bm = torch.zeros((n, h, w, 1))
  bm[:, :h // 8] = 1
  bm[:, -(h // 8):] = 1
  return dict(image=image, depth=depth, proj_mask=pm, proj_image=image * pm,
              proj_depth=depth * pm, blurred_mask=bm)


def _gin_gan(cfg, dtype):
  """The trainer exactly as the shipped gin file configures it (no bindings changed)."""
  gin_lite.clear_config()
  gin_lite.parse_config_files_and_bindings([os.path.join(ROOT, 'configs', cfg, cfg + '.gin')], [])
  gan = se3ds_trainer.GAN(strategy=gan_manager.OneDeviceStrategy(DEV), model_dir='',
                          compute_dtype=dtype)
  gan.device_init = True   # draw the 1.1 B initial values on the device (seconds, not minutes)
  gan._create_obj()
  return gan


def _oracle_cfg(gan):
  return dict(gen=dict(gen_dims=128, resnet_version='101', context_layer='convs', z_dim=128),
              dis=dict(n_dis=2, n_layers=6, kernel_size=4),
              lambda_gan=gan.lambda_gan, lambda_kld=gan.lambda_kld, lambda_wc=gan.lambda_wc,
              lambda_depth=gan.lambda_depth, mask_blurred=gan.mask_blurred,
              g_train=lambda k: not k.endswith(('/u', '/moving_mean', '/moving_variance')),
              d_train=lambda k: not k.endswith('/u'))


def _cpu_params(model):
  return {k: v.detach().cpu().clone() for k, v in model.store.views.items()}


def _capture_step(gan, batch):
  """Runs train_g_d and returns the clipped gradient arenas (as Adam consumed them) and the
  parameter arenas before the update."""
  from tests.test_nets_gpu import capture_clipped_grads
  cap = {}
  for opt, tag in ((gan.g_optimizer, 'g'), (gan.d_optimizer, 'd')):
    cap[tag + '_theta0'] = opt.model.store.theta.detach().cpu().clone()
    cap[tag + '_grad'] = capture_clipped_grads(opt)[0]
  gan.train_g_d({k: v.to(DEV) for k, v in batch.items()})
  torch.cuda.synchronize()
  return cap


def _damp_residual_branches(G, seed=9, scale=0.1):
  """A deep batch-normalised ResNet at RANDOM initialisation is numerically chaotic: every
  normalised block re-amplifies perturbations (mean-field theory of batch norm: ~1.2-1.3x per
  layer), so after the generator's 200+ layers fp32 rounding noise reaches O(1) -- two correct fp32
  implementations then disagree on activations and gradients alike (measured here: the fp32 oracle
  sits 20-30 % from its own fp64 run).  Trained networks are not in that regime.  The full-network
  parity tests therefore scale the LAST batch-norm gain of every residual block (Bottleneck.bn3,
  TransBasicBlock.bn_b) to ~0.1, which makes each block a small perturbation of the identity
  ("zero-init residual"-style weights) and the whole map well-conditioned; every other parameter
  keeps its Keras initial value.  The gin configuration is untouched."""
  gen = torch.Generator().manual_seed(seed)
  upd = {}
  for n_ in G.store.trainable_names:
    if n_.endswith(('/bn3/gamma', '/bn_b/gamma')):
      upd[n_] = (scale * (torch.rand(G.store[n_].shape, generator=gen) + 0.5)).numpy()
  assert upd
  G.store.load_dict(upd)


def _judge_against_fp64(tag, direct_ok, pairs, bad_hip, bad_orc):
  """Gate for tensors that miss 1e-3 against the fp32 oracle directly: measured against an fp64
  run of the oracle, the HIP errors must be distributed like the fp32 oracle's own errors.
  Per tensor the max-norm error of EITHER fp32 implementation is dominated by a few ReLU
  derivatives that flip for activations within rounding noise of zero (one flipped pixel of a
  4x8 map is 3 % of a channel's gradient), so single tensors are 5x off in both directions;
  what a correct implementation cannot do is be off MORE OFTEN or by MORE than the oracle.
  Gate: the error quantiles (50 / 90 / 99 %) stay within 2x of the oracle's (+1e-3), and the
  tensors where HIP is > 5x worse are not more than twice those where the oracle is > 5x worse
  (+2 % of the tensors).  No cosine fallback."""
  if not pairs:
    print(f'{tag}: all {direct_ok} tensors within 1e-3 of the fp32 oracle')
    return
  eh = np.array([a for a, _ in pairs])
  eo = np.array([b for _, b in pairs])
  qs = [50, 90, 99]
  qh, qo = np.percentile(eh, qs), np.percentile(eo, qs)
  print(f'{tag}: {direct_ok} tensors within 1e-3 of the fp32 oracle, {len(pairs)} judged against fp64: '
        f'error quantiles 50/90/99 % hip {qh[0]:.2e}/{qh[1]:.2e}/{qh[2]:.2e}, fp32 oracle '
        f'{qo[0]:.2e}/{qo[1]:.2e}/{qo[2]:.2e}; > 5x worse: hip {len(bad_hip)}, oracle {len(bad_orc)}')
  for ratio, name, e_direct, e_hip, e_o32 in sorted(bad_hip, reverse=True)[:8]:
    print(f'    {name}: direct {e_direct:.2e}; vs fp64: hip {e_hip:.2e}, fp32 oracle {e_o32:.2e}')
  for a, b, q in zip(qh, qo, qs):
    assert a <= 2.0 * b + 1e-3, (tag, q, a, b)
  assert len(bad_hip) <= 2 * len(bad_orc) + 0.02 * (len(pairs) + direct_ok), (tag, len(bad_hip), len(bad_orc))


# bf16 whole-arena gradient gates of test_cfg1_lowres_train_g_d_fp32_and_bf16 (measured values in
# the test's printout; set in round 5 from runs on MI355X with ~2x margin)
# measured (round 5, MI355X): G cosine 0.793 / ||diff|| / ||ref|| 0.645 = 32 x max(o32, 0.02) with the
# fp32 oracle 0.0055 from fp64; D cosine 0.964 / 0.267 = 13 x (fp32 oracle 0.0009 from fp64).
# Round 2 measured the same cosines (0.79 / 0.96): the figures are stable properties of the arithmetic.
BF16_CFG1_COS_MIN = {'g': 0.65, 'd': 0.90}
BF16_CFG1_K_REL = {'g': 50.0, 'd': 25.0}
# ... and of test_cfg1_generator_gradients_vs_fp64_yardstick (moving statistics, random init, 200+
# layers): measured whole-arena cosine 0.346, per-tensor cosine median 0.410 (10th percentile 0.295)
# over 888 tensors -- what 8 mantissa bits leave of that gradient ("chaotic
# amplification"); a sign or scale error in any bf16 backward kernel drives them to ~0
BF16_YARD_COS_MIN = 0.20
BF16_YARD_MEDIAN_TENSOR_COS_MIN = 0.25


def _grad_view(store, arena, name):
  o, n, shape = store._off_tr[name]
  return arena[o:o + n].view(shape)


# ======================================================================================= cfg1
def test_cfg1_lowres_train_g_d_fp32_and_bf16():
  """One train_g_d of configs/lowres/lowres.gin (128x256, batch 2, gen_dims 128, ResNet-101, 6-layer
  2-scale D) vs the oracle.  fp32: every clipped gradient tensor within 1e-3 of the oracle
  (north_star) -- a tensor that misses 1e-3 directly must be as close to an fp64 run of the oracle
  as the fp32 oracle itself is (factor 5), and the list of such tensors is printed; there is no
  cosine fallback.  bf16 (the bench's arithmetic) on the same weights: losses and the gradient
  direction track the fp32 oracle."""
  size, n = 128, 2
  batch = synth_batch(n, size, seed=4321)
  gan = _gin_gan('lowres', torch.float32)
  assert gan.image_size == 128 and gan.d_step_per_g_step == 2 and gan.mask_blurred is True
  assert gan.generator.store.theta.numel() > 1.1e9
  _damp_residual_branches(gan.generator)   # well-conditioned weights (see the helper); config untouched
  gp, dp = _cpu_params(gan.generator), _cpu_params(gan.discriminator)
  cfg = _oracle_cfg(gan)
  t0 = time.time()
  ref = O.train_g_d(gp, dp, batch, cfg)
  print(f'cfg1 oracle fp32: {time.time() - t0:.1f} s')
  cap = _capture_step(gan, batch)

  # ---- fp32 gradients, tensor by tensor.  At these dimensions (200+ layers, training-mode batch
  # statistics over as few as 64 samples per channel, random initialisation) the backward pass is
  # ill-conditioned in fp32: the fp32 ORACLE's gradients sit 20-30 % from an fp64 run of itself,
  # so no fp32 implementation (TF included) can agree with another to 1e-3 here.  The forward pass
  # is well-conditioned (losses agree to 1e-3 below).  Gate: every tensor either matches the oracle
  # to 1e-3 directly, or is as close to the fp64 run as the fp32 oracle is (factor 5) -- no cosine
  # fallback; strict per-tensor 1e-3 on the same widths is
  # test_cfg1_generator_gradients_vs_fp64_yardstick.
  torch.set_default_dtype(torch.float64)
  try:
    f64 = lambda d: {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v)
                     for k, v in d.items()}
    t0 = time.time()
    ref64 = O.train_g_d(f64(gp), f64(dp), f64(batch), cfg)
    print(f'cfg1 oracle fp64: {time.time() - t0:.1f} s')
  finally:
    torch.set_default_dtype(torch.float32)
  arena_dist = {}
  for tag, model, key in (('g', gan.generator, 'g_grads'), ('d', gan.discriminator, 'd_grads')):
    st = model.store
    assert set(ref[key]) == set(st.trainable_names)
    gmax = max(float(g.abs().max()) for g in ref64[key].values())
    direct_ok, pairs, bad_hip, bad_orc = 0, [], [], []
    for name in st.trainable_names:
      go = ref[key][name].numpy()
      gh = _grad_view(st, cap[tag + '_grad'], name).numpy()
      r64 = ref64[key][name].numpy()
      # relative to the tensor's largest entry, floored at 1e-4 of the model's largest gradient
      # entry (tensors whose true gradient is ~0, e.g. biases in front of a batch norm)
      den = max(np.abs(r64).max(), 1e-4 * gmax)
      e_direct = float(np.abs(gh - go).max() / den)
      if e_direct < 1e-3:
        direct_ok += 1
        continue
      e_hip = float(np.abs(gh - r64).max() / den)
      e_o32 = float(np.abs(go - r64).max() / den)
      pairs.append((e_hip, e_o32))
      if e_hip > 5.0 * e_o32 + 1e-3:
        bad_hip.append((e_hip / (5.0 * e_o32 + 1e-3), name, e_direct, e_hip, e_o32))
      if e_o32 > 5.0 * e_hip + 1e-3:
        bad_orc.append(name)
    _judge_against_fp64(f'cfg1 fp32 {tag}', direct_ok, pairs, bad_hip, bad_orc)
    # whole-arena distances against the fp64 run: the yardstick of the bf16 gate below
    a64 = torch.cat([ref64[key][nm].reshape(-1).double() for nm in st.trainable_names])
    a32 = torch.cat([ref[key][nm].reshape(-1).double() for nm in st.trainable_names])
    ah = torch.cat([_grad_view(st, cap[tag + '_grad'], nm).reshape(-1).double() for nm in st.trainable_names])
    arena_dist[tag] = dict(o32=float((a32 - a64).norm() / a64.norm()), hip32=float((ah - a64).norm() / a64.norm()),
                           cos_o32=float((a32 @ a64) / (a32.norm() * a64.norm())))
    print(f'cfg1 {tag}: ||g - g64|| / ||g64||: fp32 oracle {arena_dist[tag]["o32"]:.4f}, HIP fp32 '
          f'{arena_dist[tag]["hip32"]:.4f}; cosine(fp32 oracle, fp64) {arena_dist[tag]["cos_o32"]:.4f}')
    del a64, a32, ah

  # ---- Adam at t = 1 on ALL parameters (Keras form, gan_manager.py:175-183)
  for tag, opt, lr in (('g', gan.g_optimizer, gan.g_lr), ('d', gan.d_optimizer, gan.d_lr)):
    g_all, th0 = cap[tag + '_grad'], cap[tag + '_theta0']
    for o in range(0, g_all.numel(), 1 << 26):   # chunks bound the host memory
      g, p0 = g_all[o:o + (1 << 26)], th0[o:o + (1 << 26)]
      want, m1, v1 = O.adam_keras(p0, g, torch.zeros_like(g), torch.zeros_like(g), lr, gan.beta1,
                                  gan.beta2, 1)
      sl = slice(o, o + g.numel())
      step = float((p0 - want).abs().max())
      # one fp32 ulp of the parameter (the update itself must agree to 1e-5 of its size)
      assert torch.allclose(opt.model.store.theta[sl].cpu(), want, rtol=1.2e-7, atol=1e-5 * step), tag
      assert float((opt.m[sl].cpu() - m1).abs().max()) <= 1e-6 * float(m1.abs().max()) + 1e-12
      assert float((opt.v[sl].cpu() - v1).abs().max()) <= 1e-6 * float(v1.abs().max()) + 1e-12
  # ---- BN moving statistics / spectral u after the step
  bad = []
  for k, v in ref['g_updates'].items():
    e = rel_err(gan.generator.store[k].cpu().numpy(), v.detach().numpy())
    if e >= 1e-3:
      bad.append((k, e))
  assert not bad, bad[:5]
  # ---- EMA: hard copy in the first cluster (gan_manager.py:642-655)
  assert torch.equal(gan.ema_generator.store.theta, gan.generator.store.theta)
  m32 = gan._save_metrics_to_dict()
  for key in ('gen/gen_gan_loss', 'dis/disc_loss', 'gen/depth_loss', 'gen/wc_loss', 'gen/gen_loss'):
    r = ref['metrics'][key]   # forward quantities: well-conditioned, direct 1e-3
    assert abs(float(m32[key]) - r) <= 1e-3 * max(1.0, abs(r)), (key, float(m32[key]), r)
  for key in ('gen/grad_norm', 'dis/grad_norm'):   # functions of the gradients: fp64 yardstick
    r64, r32 = ref64['metrics'][key], ref['metrics'][key]
    assert abs(float(m32[key]) - r64) <= 5.0 * abs(r32 - r64) + 1e-3 * max(1.0, abs(r64)), \
        (key, float(m32[key]), r32, r64)
  del ref64

  # ---- bf16 path on the same initial weights
  theta_g, state_g = cap['g_theta0'], {k: gp[k] for k in gan.generator.store.state_names}
  theta_d, state_d = cap['d_theta0'], {k: dp[k] for k in gan.discriminator.store.state_names}
  del gan
  torch.cuda.empty_cache()
  gan = _gin_gan('lowres', torch.bfloat16)
  for model, theta, state in ((gan.generator, theta_g, state_g), (gan.discriminator, theta_d, state_d)):
    model.store.theta.copy_(theta.to(DEV))
    model.store.load_dict({k: v.numpy() for k, v in state.items()})
  cap16 = _capture_step(gan, batch)
  m16 = gan._save_metrics_to_dict()
  for key in ('gen/gen_gan_loss', 'dis/disc_loss', 'gen/depth_loss', 'gen/wc_loss'):
    r = ref['metrics'][key]
    print(f'cfg1 bf16 {key}: {float(m16[key]):.5f} (oracle fp32 {r:.5f})')
    # 8 mantissa bits through 200+ layers with batch statistics over 64 samples: the mean patch
    # logit (a difference of O(1) terms) moves by a few 1e-2; the L1 losses by a few 1e-3
    assert abs(float(m16[key]) - r) <= (1e-1 if 'gan' in key or 'disc' in key else 2e-2) * max(1.0, abs(r)), \
        (key, float(m16[key]), r)
  for tag, model, key in (('g', gan.generator, 'g_grads'), ('d', gan.discriminator, 'd_grads')):
    st = model.store
    a = cap16[tag + '_grad'].double()
    b = torch.zeros_like(a)
    for name in st.trainable_names:
      o, cnt, _ = st._off_tr[name]
      b[o:o + cnt] = ref[key][name].reshape(-1).double()
    cos = float((a @ b) / (a.norm() * b.norm()))
    rel = float((a - b).norm() / b.norm())
    # Round 5: GATED against this test's own yardstick.  At random initialisation with batch
    # statistics over 64 samples the backward pass is ill-conditioned: the fp32 oracle's whole-arena
    # gradient sits `o32` from an fp64 run of itself (printed above).  bf16 rounds every activation
    # to 8 mantissa bits (2^-9 relative against fp32's 2^-24), so its distance from the fp32
    # oracle may exceed that fp32 noise floor by a bounded factor, and the direction must hold:
    #   ||g_bf16 - g_o32|| / ||g_o32||  <=  K_REL x max(o32, 0.02)      cosine >= COS_MIN
    # (measured round 5 on MI355X: see the printed line; K_REL and COS_MIN leave ~2x margin).  The
    # bf16 kernels are additionally gated layer by layer (tests/test_prod_shapes_gpu.py), block by
    # block (tests/test_blocks_gpu.py) and as a training signal (the trajectory test below).
    o32 = arena_dist[tag]['o32']
    print(f'cfg1 bf16 {tag}: gradient cosine vs fp32 oracle {cos:.4f}, ||diff||/||ref|| {rel:.3f} '
          f'= {rel / max(o32, 0.02):.1f} x the fp32 oracle\'s distance from fp64 ({o32:.4f})')
    assert cos >= BF16_CFG1_COS_MIN[tag], (tag, cos)
    assert rel <= BF16_CFG1_K_REL[tag] * max(o32, 0.02), (tag, rel, o32)
    assert bool(torch.isfinite(cap16[tag + '_grad']).all())
  assert bool(torch.isfinite(gan.generator.store.theta).all())


def test_cfg1_generator_gradients_vs_fp64_yardstick():
  """Every generator parameter gradient at cfg1's real width and depth (gen_dims 128, ResNet-101;
  64x128 panorama, batch 1 -- at 32x64 the 1x2 bottleneck map saturates both heads) vs oracle autograd: zero padding (training flag) but batch norm on
  (randomised, calibrated) moving statistics, random affine / bias values, random cotangents on
  rgb and depth.  Without batch statistics the map is far better conditioned than a training
  step, but 200+ layers of fp32 rounding and ReLU kinks still leave most tensors outside a direct
  1e-3 of the fp32 oracle: those are held to the fp64 yardstick (as close to an fp64 run as the
  fp32 oracle itself is).  The DIRECT 1e-3 per-tensor bar at production width lives in
  tests/test_blocks_gpu.py, where a block is short enough for it."""
  gin_lite.clear_config()
  G = image_models.ResNetGenerator(image_size=64, gen_dims=128, resnet_version='101', device=DEV,
                                   seed=-3, dtype=torch.float32)
  batch = synth_batch(1, 64, seed=55)
  _randomise_inference_state(G, batch)
  names = G.store.trainable_names
  snap = _cpu_params(G)   # ONE snapshot: the training-flag forward below advances the spectral `u`
  def oracle(dt):
    p = {k: v.to(dt if v.is_floating_point() else v.dtype).clone().requires_grad_(k in names)
         for k, v in snap.items()}
    b = {k: v.to(dt) for k, v in batch.items()}
    outs_o, _ = O.generator_forward(p, b, True, gen_dims=128, resnet_version='101', z_dim=128,
                                    bn_training=False)
    ((outs_o[6] * w_rgb.to(dt)).sum() + (outs_o[3] * w_d.to(dt)).sum()).backward()
    return outs_o, {k: p[k].grad for k in names}
  gen = torch.Generator().manual_seed(6)
  w_rgb = torch.randn((1, 64, 128, 3), generator=gen)
  w_d = torch.randn((1, 64, 128, 1), generator=gen)
  t0 = time.time()
  outs_o, g32 = oracle(torch.float32)
  print(f'oracle generator fwd+bwd fp32: {time.time() - t0:.1f} s')
  ctx = G.make_ctx(True, record=True)
  ctx.bn_use_moving = True
  outs, (push_rgb, push_depth) = G.forward(ctx, {k: v.to(DEV) for k, v in batch.items()})
  assert rel_err(outs[6].cpu().numpy(), outs_o[6].detach().numpy()) < 1e-4
  assert rel_err(outs[3].cpu().numpy(), outs_o[3].detach().numpy()) < 1e-4
  push_rgb(w_rgb.to(DEV))
  push_depth(w_d.to(DEV))
  ctx.backward()
  G.spectral.backward_fixup()
  gmax = max(float(g32[k].abs().max()) for k in names)
  def err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-4 * gmax))
  gh = {k: G.store.grad_views[k].cpu().numpy() for k in names}
  direct = {k: err(gh[k], g32[k].numpy()) for k in names}
  misses = [k for k in names if direct[k] >= 1e-3]
  print(f'{len(names)} tensors: max {max(direct.values()):.2e}, median {np.median(list(direct.values())):.2e}, '
        f'{len(misses)} above 1e-3, {len(names) - len(misses)} within 1e-3 of the fp32 oracle DIRECTLY')
  # the count of tensors that meet north_star's 1e-3 bar directly is gated, so that a regression in
  # direct agreement is visible even while the fp64 yardstick below still passes (measured in
  # round 4: 74 of 954 -- at cfg1's depth two correct fp32 evaluations differ by 1e-2 on the rest)
  assert len(names) - len(misses) >= 60, (len(names), len(misses))
  if misses:
    # 200+ layers of fp32 rounding and ReLU kinks (an activation within fp32 noise of zero flips its
    # derivative between two correct implementations): judge those tensors by the fp64 yardstick (as accurate as the fp32 oracle, factor 5); still no cosine fallback
    torch.set_default_dtype(torch.float64)
    try:
      t0 = time.time()
      _, g64 = oracle(torch.float64)
      print(f'oracle generator fwd+bwd fp64: {time.time() - t0:.1f} s')
    finally:
      torch.set_default_dtype(torch.float32)
    pairs, bad_hip, bad_orc = [], [], []
    for k in misses:
      r64 = g64[k].numpy()
      e_hip, e_o32 = err(gh[k], r64), err(g32[k].numpy(), r64)
      pairs.append((e_hip, e_o32))
      if e_hip > 5.0 * e_o32 + 1e-3:
        bad_hip.append((e_hip / (5.0 * e_o32 + 1e-3), k, direct[k], e_hip, e_o32))
      if e_o32 > 5.0 * e_hip + 1e-3:
        bad_orc.append(k)
    _judge_against_fp64('generator (moving statistics)', len(names) - len(misses), pairs, bad_hip, bad_orc)
  # ---- the bf16 path on the same (well-conditioned) network: direction of the whole gradient
  Gb = image_models.ResNetGenerator(image_size=64, gen_dims=128, resnet_version='101', device=DEV,
                                    seed=-3, dtype=torch.bfloat16)
  Gb.store.load_dict({k: v.numpy() for k, v in snap.items()})
  ctx = Gb.make_ctx(True, record=True)
  ctx.bn_use_moving = True
  outs, (push_rgb, push_depth) = Gb.forward(ctx, {k: v.to(DEV) for k, v in batch.items()})
  push_rgb(w_rgb.to(DEV))
  push_depth(w_d.to(DEV))
  ctx.backward()
  Gb.spectral.backward_fixup()
  a = Gb.store.grad.double().cpu()
  bref = torch.zeros_like(a)
  for k in names:
    o, cnt, _ = Gb.store._off_tr[k]
    bref[o:o + cnt] = g32[k].reshape(-1).double()
  cos = float((a @ bref) / (a.norm() * bref.norm()))
  # per-tensor cosines (tensors whose reference gradient is numerically zero are skipped)
  tcos = []
  floor = 1e-6 * float(bref.norm()) / max(len(names), 1) ** 0.5
  for k in names:
    o, cnt, _ = Gb.store._off_tr[k]
    x, y = a[o:o + cnt], bref[o:o + cnt]
    if float(y.norm()) > floor:
      tcos.append(float((x @ y) / (x.norm() * y.norm() + 1e-300)))
  tmed, t10 = float(np.median(tcos)), float(np.percentile(tcos, 10))
  print(f'bf16 path: gradient cosine vs fp32 oracle {cos:.4f}, ||diff||/||ref|| '
        f'{float((a - bref).norm() / bref.norm()):.3f}; per-tensor cosine median {tmed:.4f}, 10th percentile '
        f'{t10:.4f} over {len(tcos)} tensors; rgb max err {rel_err(outs[6].cpu().numpy(), outs_o[6].detach().numpy()):.2e}')
  # Round 5: gated (was `cos > 0`): the measured values minus a margin -- what 8 mantissa bits leave
  # of a 200-layer gradient on the well-conditioned (moving-statistics) network
  assert cos >= BF16_YARD_COS_MIN, cos
  assert tmed >= BF16_YARD_MEDIAN_TENSOR_COS_MIN, tmed


def test_cfg1_bf16_training_trajectory_tracks_fp32():
  """End-to-end gate of the bf16 step, the arithmetic of the headline number (the reference trains
  in fp32; per-layer and per-block bf16 parity is pinned in test_prod_shapes_gpu / test_blocks_gpu).
  30 `train_g_d` steps (full lowres.gin trainer: ResNet-101 gen_dims 128, 128x256, batch 2, Adam
  on all 1.1 B parameters, EMA) on ONE fixed batch, once on the fp32 path and once on the bf16
  path, from identical initial weights.  Gates:
    * the supervised part of the generator loss (depth L1 + world-consistency) falls on BOTH paths:
      mean of the last 5 steps < 0.8 x mean of the first 3;
    * the bf16 trajectory stays inside a band around the fp32 one: the 5-step moving average of
      depth+wc within 15 % (+0.05 abs) or within 3x the deviation a 1e-6 perturbation of the fp32
      run's own weights causes (the noise yardstick, measured in the same test), the 30-step mean
      deviation within 6 %, and the 30-step means of depth, wc and the discriminator loss within 10 %;
    * everything stays finite.
  A bf16 step that returned gradients of the wrong sign or scale (the per-tensor cosine of a
  200-layer gradient at random init is 0.3-0.8: what 8 mantissa bits leave of a chaotic map) would
  show here as a loss that stalls or leaves the band."""
  steps = 30
  batch = synth_batch(2, 128, seed=91)
  dbatch = {k: v.to(DEV) for k, v in batch.items()}
  traj = {}
  theta0 = None
  # third run: fp32 again from weights perturbed by 1e-6 relative -- the trajectory's own
  # sensitivity to rounding-level noise, the yardstick for the per-step band below (round 4: pinning
  # the optimiser's fused multiply-adds moved the bf16-fp32 gap of one step from 9 % to 19 % while
  # both losses kept falling together: the map is chaotic at that level, a fixed band is not a
  # property of the arithmetic)
  for tag, dtype in (('fp32', torch.float32), ('bf16', torch.bfloat16), ('fp32+1e-6', torch.float32)):
    gan = _gin_gan('lowres', dtype)
    if theta0 is None:
      theta0 = [m.store.theta.clone() for m in (gan.generator, gan.discriminator)]
      state0 = [m.store.state.clone() for m in (gan.generator, gan.discriminator)]
    else:   # identical start (device-side init is seeded, this makes it explicit)
      for m, t, st in zip((gan.generator, gan.discriminator), theta0, state0):
        m.store.theta.copy_(t)
        if tag == 'fp32+1e-6':
          g = torch.Generator(device=DEV).manual_seed(7)
          m.store.theta.mul_(1.0 + 1e-6 * torch.randn(t.shape, generator=g, device=DEV))
        m.store.state.copy_(st)
        m.store.version += 1
    rows = []
    t0 = time.time()
    for _ in range(steps):
      gan._reset_metrics()
      gan.train_g_d(dbatch)
      gan.global_step += gan.num_batched_steps
      m = gan._save_metrics_to_dict()
      rows.append([float(m[k]) for k in ('gen/depth_loss', 'gen/wc_loss', 'dis/disc_loss',
                                          'gen/gen_gan_loss')])
    torch.cuda.synchronize()
    traj[tag] = np.array(rows, np.float64)
    assert np.isfinite(traj[tag]).all() and bool(torch.isfinite(gan.generator.store.theta).all())
    print(f'{tag}: {steps} steps in {time.time() - t0:.1f} s; depth+wc '
          + ' '.join(f'{v:.3f}' for v in (traj[tag][:, 0] + traj[tag][:, 1])[::3]))
    del gan
    torch.cuda.empty_cache()
  traj[torch.float32], traj[torch.bfloat16] = traj['fp32'], traj['bf16']
  f, b = traj[torch.float32], traj[torch.bfloat16]
  sup_f, sup_b = f[:, 0] + f[:, 1], b[:, 0] + b[:, 1]
  for tag, sup in (('fp32', sup_f), ('bf16', sup_b)):
    assert sup[-5:].mean() < 0.8 * sup[:3].mean(), (tag, sup[:3].mean(), sup[-5:].mean())
  dev = np.abs(sup_b - sup_f) / (np.abs(sup_f) + 1e-12)
  p = traj['fp32+1e-6']
  sup_p = p[:, 0] + p[:, 1]
  noise = np.abs(sup_p - sup_f) / (np.abs(sup_f) + 1e-12)
  print(f'bf16 vs fp32 depth+wc: max per-step deviation {dev.max():.3f} (step {int(dev.argmax())}), '
        f'mean {dev.mean():.3f}; fp32 vs fp32 perturbed by 1e-6: max {noise.max():.3f}, mean {noise.mean():.3f}; '
        f'disc means {f[:, 2].mean():.4f} / {b[:, 2].mean():.4f}')
  # The per-step values of a batch-2 GAN trajectory SPIKE (round 4: the fp32 run itself jumps by
  # 2.5 % upwards around step 15-20 where the perturbed fp32 run and the bf16 run keep falling; the
  # bf16 run is closer to the perturbed fp32 run than the fp32 run is).  The band is therefore
  # applied to 5-step moving averages: inside 15 % -- or inside 3x what the 1e-6 perturbation does
  # to the fp32 trajectory's moving average (running maximum: deviations grow along a chaotic
  # trajectory); and on average (30 steps, unsmoothed) inside 6 % or 3x the noise.
  ma = lambda v: np.convolve(v, np.ones(5) / 5.0, mode='valid')
  ma_f, ma_b, ma_p = ma(sup_f), ma(sup_b), ma(sup_p)
  ma_noise = np.abs(ma_p - ma_f) / ma_f
  ma_dev = np.abs(ma_b - ma_f) / ma_f
  print(f'5-step moving averages: bf16 vs fp32 max {ma_dev.max():.3f}, fp32 vs perturbed fp32 max {ma_noise.max():.3f}')
  band = np.maximum(0.15, 3.0 * np.maximum.accumulate(ma_noise))
  assert (np.abs(ma_b - ma_f) <= band * ma_f + 0.05).all(), (ma_dev, ma_noise)
  assert dev.mean() <= max(0.06, 3.0 * noise.mean()), (dev.mean(), noise.mean())
  for col, name in ((0, 'depth'), (1, 'wc'), (2, 'disc')):
    mf, mb = f[:, col].mean(), b[:, col].mean()
    assert abs(mb - mf) <= 0.10 * abs(mf) + 1e-3, (name, mf, mb)


# ======================================================================================= cfg2
def _randomise_inference_state(G, batch=None, seed=4):
  """Non-trivial inference state: random affine values / biases, and batch-norm moving statistics
  CALIBRATED to the network's own activations on `batch` (a dict of CPU tensors; default: a
  synthetic batch at the generator's working size).  At initialisation the moving statistics are
  (0, 1): nothing is normalised, every residual block grows the activations, and after 100+
  layers tanh / clip saturate -- outputs become trivial and gradients meaningless.  Calibration =
  one training-mode forward on the device; the batch statistics are recovered from the momentum
  update moving' = 0.99 moving + 0.01 batch (exactness is irrelevant: the oracle gets the same
  numbers)."""
  gen = torch.Generator().manual_seed(seed)
  upd = {}
  for n_ in G.store.trainable_names:
    if n_.endswith('gamma'):
      upd[n_] = (torch.rand(G.store[n_].shape, generator=gen) * 0.2 + 0.9).numpy()
    if n_.endswith('beta') or n_.endswith('bias'):
      upd[n_] = (torch.randn(G.store[n_].shape, generator=gen) * 0.1).numpy()
  G.store.load_dict(upd)
  _damp_residual_branches(G, seed + 1)
  if batch is None:
    batch = synth_batch(1, 64, seed=seed)
  stats = [n_ for n_ in G.store.state_names if n_.endswith(('moving_mean', 'moving_variance'))]
  before = {n_: G.store[n_].clone() for n_ in stats}
  u0 = {n_: G.store[n_].clone() for n_ in G.store.state_names if n_.endswith('/u')}
  G.forward(G.make_ctx(True), {k: v.to(DEV) for k, v in batch.items()})
  for n_ in stats:
    est = (G.store[n_] - 0.99 * before[n_]) / 0.01
    if n_.endswith('moving_variance'):
      est = torch.clamp(est, min=1e-3)
    G.store[n_].copy_(est)
  for n_, v in u0.items():   # the calibration pass must not count as a power-iteration step
    G.store[n_].copy_(v)
  G.store.version += 1


def _check_frame(out, ref, tag):
  """Integer / index outputs of the warp half bit-exact; generator outputs within 1e-3."""
  np.testing.assert_array_equal(out.proj_semantic.cpu().numpy(), ref['proj_semantic'], tag)
  np.testing.assert_array_equal(out.proj_rgb.cpu().numpy(), ref['proj_rgb'], tag)
  np.testing.assert_array_equal(out.proj_depth.cpu().numpy(), ref['proj_depth'], tag)
  np.testing.assert_array_equal(out.proj_mask.cpu().numpy(), ref['proj_mask'], tag)
  e_d = rel_err(out.pred_depth.cpu().numpy(), ref['pred_depth'])
  assert e_d < 1e-3, (tag, e_d)
  a, b = out.pred_rgb.cpu().numpy().astype(np.int32), ref['pred_rgb'].astype(np.int32)
  assert out.pred_rgb.dtype == torch.uint8 and a.shape == b.shape
  # truncation of g * 255: an fp32-noise difference in g flips a value only next to an integer
  assert np.abs(a - b).max() <= 1 and np.mean(a != b) < 2e-3, (tag, np.mean(a != b))
  np.testing.assert_array_equal(out.pred_semantic.cpu().numpy(), ref['pred_semantic'])
  return e_d


def test_cfg2_inference_256x512_fp32_with_warp():
  size = 256
  gin_lite.clear_config()
  config = model_config.get_config()
  config.ckpt_path = None
  config.image_height = size
  assert config.gen_dims == 128 and config.resnet_version == '101'
  model = models.SE3DSModel(config, device=DEV, dtype=torch.float32)
  _randomise_inference_state(model.model)
  rng = np.random.default_rng(5)
  frames = []
  for _ in range(2):
    rgb = rng.integers(0, 256, (1, size, 2 * size, 3)).astype(np.uint8)
    seg = rng.integers(0, 42, (1, size, 2 * size, 1)).astype(np.uint8)
    depth = rng.uniform(0, 1, (1, size, 2 * size)).astype(F32)
    poison = rng.uniform(0, 1, depth.shape)
    depth[poison < 0.02] = 0
    depth[poison > 0.99] = 1
    pos = (rng.standard_normal((1, 3)) * 0.5).astype(F32)
    frames.append((rgb, seg, depth, pos))
  oracle = model_np.SE3DSModelOracle(_cpu_params(model.model), size, 128, '101')
  t = lambda a: torch.from_numpy(a).to(DEV)
  for rgb, seg, depth, pos in frames:
    model.add_to_memory(t(rgb), t(seg), t(depth), t(pos))
    oracle.add_to_memory(rgb, seg, depth, pos)
  def check_memory(tag):
    ms = model.get_memory_state()
    np.testing.assert_array_equal(ms.coords.cpu().numpy(), oracle.coords, tag)
    np.testing.assert_array_equal(ms.feats.cpu().numpy(), oracle.feats, tag)
    np.testing.assert_array_equal(ms.rgb_coords.cpu().numpy(), oracle.rgb_coords, tag)
    np.testing.assert_array_equal(ms.rgb.cpu().numpy(), oracle.rgb, tag)
  check_memory('after add_to_memory')
  # the RGB memory's [-1, 255] bounds are known by construction along quantise -> mask_pano ->
  # unproject -> compact -> concat -> copy: the packed splat's byte-range promise needs no device
  # read-back (one blocking .item() per frame otherwise)
  from se3ds_amd.utils import point_cloud_utils
  assert point_cloud_utils.get_int_range(model._memory.rgb) == (-1, 255)
  # 2 feedback frames + 1 plain call (gan_manager.py:458-556's loop body; models.py:334-346; the
  # 4-frame roll-out proper is test_autoregressive_rollout_vs_oracle).
  # After every frame the oracle's memory is re-synchronised to the HIP memory: the fed-back
  # integers come out of a float network, so they agree only up to +-1 on a few pixels.
  for i in range(2):
    target = (rng.standard_normal((1, 3)) * 0.5).astype(F32)
    out = model(t(target), add_preds_to_memory=True)
    ref = oracle(target, add_preds_to_memory=True)
    e = _check_frame(out, ref, f'frame {i}')
    print(f'cfg2 frame {i}: pred_depth err {e:.2e}, memory {model.get_memory_state().rgb.shape[1]} points')
    ms = model.get_memory_state()
    assert ms.rgb.shape[1] >= oracle.rgb.shape[1] - 200 and ms.rgb.dtype == torch.int32
    oracle.coords, oracle.feats = ms.coords.cpu().numpy(), ms.feats.cpu().numpy()
    oracle.rgb_coords, oracle.rgb = ms.rgb_coords.cpu().numpy(), ms.rgb.cpu().numpy()
  # plain call (no feedback) after the roll-out
  target = (rng.standard_normal((1, 3)) * 0.5).astype(F32)
  _check_frame(model(t(target)), oracle(target), 'final')


# ======================================================================================= cfg3
def test_cfg3_highres_bf16_step_properties():
  """configs/highres/highres.gin, 512x1024, bf16: a batch-1 train_g_d's loss values vs the
  oracle's fp32 forward on the same weights, finite state after the update; then a batch-2 step
  (the shape class the bench runs) stays finite."""
  size = 512
  gan = _gin_gan('highres', torch.bfloat16)
  assert gan.image_size == 512
  batch = synth_batch(1, size, seed=99)
  gp, dp = _cpu_params(gan.generator), _cpu_params(gan.discriminator)
  cfg = _oracle_cfg(gan)
  t0 = time.time()
  with torch.no_grad():
    inputs = dict(batch)
    outs, _ = O.generator_forward(gp, inputs, True, **cfg['gen'])
    depth_out, generated = outs[3], outs[6]
    depth_t = batch['depth']
    tmask = ((depth_t > 0) & (depth_t < 1)).float()
    depth_loss = cfg['lambda_depth'] * ((torch.abs(depth_out - depth_t) * tmask).sum(dim=(1, 2, 3)) /
                                        torch.clamp(tmask.sum(dim=(1, 2, 3)), min=1)).mean()
    wc = cfg['lambda_wc'] * O.wc_loss(generated, batch['proj_image'],
                                      batch['proj_mask'] * (1 - batch['blurred_mask'])).mean()
    fake = torch.cat([generated, depth_out], dim=-1)
    real = torch.cat([batch['image'], depth_t], dim=-1)
    logits, _ = O.discriminator_forward(dp, torch.cat([fake, real], dim=0), True, **cfg['dis'])
    gen_loss, disc_loss = O.d_losses(logits)
  print(f'cfg3 oracle forward: {time.time() - t0:.1f} s')
  want = {'gen/gen_gan_loss': float(gen_loss), 'dis/disc_loss': float(disc_loss),
          'gen/depth_loss': float(depth_loss), 'gen/wc_loss': float(wc)}
  gan.train_g_d({k: v.to(DEV) for k, v in batch.items()})
  m = gan._save_metrics_to_dict()
  for k, r in want.items():
    print(f'cfg3 bf16 {k}: {float(m[k]):.5f} (oracle fp32 {r:.5f})')
    assert abs(float(m[k]) - r) <= 3e-2 * max(1.0, abs(r)), (k, float(m[k]), r)
  assert bool(torch.isfinite(gan.generator.store.theta).all())
  assert bool(torch.isfinite(gan.discriminator.store.theta).all())
  gan._reset_metrics()
  gan.global_step += 1
  gan.train_g_d({k: v.to(DEV) for k, v in synth_batch(2, size, seed=100).items()})
  m = gan._save_metrics_to_dict()   # raises on NaN (gan_manager.py:637-638)
  assert all(np.isfinite(float(v)) for v in m.values())
  assert bool(torch.isfinite(gan.generator.store.theta).all())


def test_cfg3_fullsize_fp32_forward_vs_oracle():
  """The BENCHMARKED configuration's tensors, not only its loss scalars:
  configs/highres/highres.gin at 512x1024 on the fp32 path -- generator forward in inference mode
  on a calibrated state (as cfg2 does at 256x512), then the multi-scale discriminator on
  [fake; real]: `rgb`, `depth` (N,512,1024,.) and both logit maps (18x34 and 10x18; their
  257x513 / 129x257 / 65x129 / 33x65 / 17x33 predecessors are where tile-edge bugs would show)
  within 1e-3 of oracle/nets_torch.py (image_models.py:132-193, :599-618)."""
  size = 512
  gan = _gin_gan('highres', torch.float32)
  G, D = gan.generator, gan.discriminator
  assert gan.image_size == 512 and G.hidden_dims == 128 and G.resnet_version == '101'
  batch = synth_batch(1, size, seed=321)
  _randomise_inference_state(G, batch=batch, seed=6)
  dev_batch = {k: v.to(DEV) for k, v in batch.items()}
  outs, _ = G.forward(G.make_ctx(False), dev_batch)
  depth_g, rgb_g = outs[3], outs[6]
  assert tuple(rgb_g.shape) == (1, size, 2 * size, 3) and tuple(depth_g.shape) == (1, size, 2 * size, 1)
  ctx_d = D.make_ctx(False)
  x_all = gan._disc_input(ctx_d, rgb_g, depth_g, dev_batch['image'], dev_batch['depth'])
  logits_g = D.forward(ctx_d, x_all)
  torch.cuda.synchronize()
  cfg = _oracle_cfg(gan)
  t0 = time.time()
  with torch.no_grad():
    gp, dp = _cpu_params(G), _cpu_params(D)
    outs_o, _ = O.generator_forward(gp, dict(batch), False, **cfg['gen'])
    fake = torch.cat([outs_o[6], outs_o[3]], dim=-1)
    real = torch.cat([batch['image'], batch['depth']], dim=-1)
    logits_o, _ = O.discriminator_forward(dp, torch.cat([fake, real], dim=0), False, **cfg['dis'])
  print(f'cfg3 oracle forward (G + D, fp32, 512x1024): {time.time() - t0:.1f} s')
  e_rgb = rel_err(rgb_g.cpu().numpy(), outs_o[6].numpy())
  e_depth = rel_err(depth_g.cpu().numpy(), outs_o[3].numpy())
  print(f'cfg3 fp32 forward: rgb err {e_rgb:.2e}, depth err {e_depth:.2e}; depth range '
        f'{float(outs_o[3].min()):.3f}..{float(outs_o[3].max()):.3f}, rgb std {float(outs_o[6].std()):.3f}')
  assert float(outs_o[6].std()) > 1e-3 and float(outs_o[3].max() - outs_o[3].min()) > 1e-3, 'trivial outputs'
  assert e_rgb < 1e-3 and e_depth < 1e-3, (e_rgb, e_depth)
  want_shapes = [(2, 18, 34, 1), (2, 10, 18, 1)]
  for k, (sub_g, sub_o) in enumerate(zip(logits_g, logits_o)):
    assert len(sub_g) == len(sub_o) == 7
    for lvl, (a, b) in enumerate(zip(sub_g, sub_o)):   # every returned feature map, not only the logits
      e = rel_err(a.data.float().cpu().numpy(), b.numpy())
      assert e < 1e-3, (k, lvl, tuple(b.shape), e)
    assert tuple(sub_o[-1].shape) == want_shapes[k], tuple(sub_o[-1].shape)
    print(f'cfg3 fp32 discriminator {k}: maps {[tuple(t.shape[1:3]) for t in sub_o]}, logit err '
          f'{rel_err(sub_g[-1].data.float().cpu().numpy(), sub_o[-1].numpy()):.2e}')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_resstack_masking_on_hip_path(dtype):
  """layers_test.py:64-86 on the product's ResStack at the generator's stack-1 dimensions
  (128 -> 512 channels, 3 bottlenecks, 128x256 maps, training-mode batch statistics): changing a
  pixel inside the masked region leaves every output bit unchanged (assertAllEqual)."""
  store = nn.ParamStore()
  stack = layers.ResStack(store, 's', 128, 128, blocks=3, strides=1, conv_fn=layers.SpectralConv)
  store.finalize(DEV, torch.Generator().manual_seed(1))
  sg = nn.SpectralGroup(image_models._conv_layers_of(stack), torch.device(DEV))
  n, h, w = 2, 128, 256
  x = torch.rand((n, h, w, 128), generator=torch.Generator().manual_seed(2))
  mask = (torch.arange(h, dtype=torch.float32) > h // 2).float()[None, :, None].repeat(n, 1, w)
  x2 = x.clone()
  x2[:, 0, 0, :] = 1
  outs = []
  for xin in (x, x2):
    ctx = nn.Ctx(DEV, dtype, training=True, record=False)
    sg.power_iteration(training=False)
    y, um = stack(ctx, nn.Var(xin.to(DEV).to(dtype), requires_grad=False), mask.to(DEV).contiguous())
    outs.append((y.data.clone(), um.clone()))
  assert tuple(outs[0][0].shape) == (n, h, w, 512) and tuple(outs[0][1].shape) == (n, h, w)
  assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
  assert float(outs[0][0].float().abs().max()) > 0


# ======================================================================================= cfg5
@pytest.mark.parametrize('depth_kind', ['random', 'room'])
def test_cfg5_warp_1024x2048_two_views_bit_exact(depth_kind):
  """1024x2048, V = 2 source views (4.2 M points) rendered at a third position: unproject,
  project + splat, mask -- bit-exact vs oracle/warp_oracle.c (the C twin pinned against the
  reference's golden vectors in tests/test_oracle_warp.py) AND vs oracle/warp_np.py (reference:
  utils/point_cloud_utils.py:90-183, utils/pano_utils.py:117-161).  At this size the default
  dispatch is the round-4 sorted-chunk splat."""
  import bench
  h, w, views = 1024, 2048, 2
  rng = np.random.default_rng(77)
  panos, target = bench._warp_inputs(rng, h, w, views, DEV, depth_kind)
  tabs = warp_np.equirect_angle_tables(h, w)
  t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
  xs_o, fs_o, xs_g, fs_g = [], [], [], []
  for rgb, depth, pos in panos:
    xo, fo = warp_c.unproject_equirect(rgb, depth, tabs, -1, 20.0, position=pos)
    xg, fg = pano_utils.equirectangular_to_pointcloud(t(rgb), t(depth), -1, 20.0, position=t(pos))
    np.testing.assert_array_equal(xg.cpu().numpy(), xo)
    np.testing.assert_array_equal(fg.cpu().numpy().astype(F32), fo)
    xs_o.append(xo); fs_o.append(fo); xs_g.append(xg); fs_g.append(fg)
  mem_x, mem_f = np.concatenate(xs_o, 2), np.concatenate(fs_o, 1)
  d_o, f_o = warp_c.project_feats_to_equirectangular(mem_f, mem_x, h, w, -1, 20.0, offset=target)
  # the views are unprojected straight into their windows of one memory (no concat copy)
  gx = torch.empty((1, 4, views * h * w), dtype=torch.float32, device=DEV)
  gf = torch.empty((1, views * h * w, 3), dtype=torch.int32, device=DEV)
  for v, (rgb, depth, pos) in enumerate(panos):
    pano_utils.equirectangular_to_pointcloud(t(rgb), t(depth), -1, 20.0, position=t(pos),
                                             out=(gx, gf, v * h * w))
  assert torch.equal(gx, torch.cat(xs_g, 2)) and torch.equal(gf, torch.cat(fs_g, 1))
  d_g, f_g, m_g = pano_utils.project_feats_to_equirectangular(
      gf, gx, h, w, -1, 20.0, offset=t(target), with_mask=True)
  np.testing.assert_array_equal(d_g.cpu().numpy(), d_o)
  np.testing.assert_array_equal(f_g.cpu().numpy(), f_o)
  np.testing.assert_array_equal(m_g.cpu().numpy()[..., None], warp_np.proj_mask(d_o, f_o, -1))
  assert 0.2 < float(m_g.mean()) <= 1.0
  # ... and against the INDEPENDENT NumPy / libm statement (oracle/warp_np.py shares no arithmetic
  # with the kernels: np.arctan2 / np.arccos in binary64 rounded to fp32, NumPy scatter-min / max),
  # for BOTH depth kinds: the smooth `room` input is the one with heavy tiles (banded resolve), so
  # the largest configuration is never checked twin-vs-twin only
  for (rgb, depth, pos), xg, fg in zip(panos, xs_g, fs_g):
    xn, fn = warp_np.equirectangular_to_pointcloud(rgb, depth, -1, 20.0)
    xn = (xn + np.concatenate([pos, np.zeros((1, 1), F32)], 1)[:, :, None]).astype(F32)
    np.testing.assert_array_equal(xg.cpu().numpy(), xn)
    np.testing.assert_array_equal(fg.cpu().numpy(), fn)
  rel = (mem_x - np.concatenate([target, np.zeros((1, 1), F32)], 1)[:, :, None]).astype(F32)
  d_n, f_n = warp_np.project_feats_to_equirectangular(mem_f, rel, h, w, -1, 20.0)
  np.testing.assert_array_equal(d_g.cpu().numpy(), d_n)
  np.testing.assert_array_equal(f_g.cpu().numpy(), f_n)


def test_quantize_steps_bit_exact():
  """models.py:198,:289-291,:325-331,:353 through se3ds_quantize vs the NumPy statement order."""
  rng = np.random.default_rng(3)
  g = rng.uniform(-0.01, 1.01, (1, 64, 128, 3)).astype(F32)
  g.reshape(-1)[:512] = (np.arange(512) / 255.0).astype(F32)[:512]   # values on the integer grid
  g.reshape(-1)[512:520] = [0.0, 1.0, -0.0, 0.999999, 1.0000001, -1e-9, 0.5, 1 / 255]
  t = torch.from_numpy(g).to(DEV)
  q = models._quantize
  np.testing.assert_array_equal(
      q(t, torch.int32, mul=255.0, lo=-1, hi=255).cpu().numpy(),
      np.clip(np.trunc(g * F32(255)).astype(np.int32), -1, 255))
  np.testing.assert_array_equal(
      q(t, torch.int32, mul=255.0, lo=0, hi=255, pre=(0.0, 1.0)).cpu().numpy(),
      np.trunc(np.clip(g, 0, 1) * F32(255)).astype(np.int32))
  np.testing.assert_array_equal(q(t[..., 0], torch.float32, lo=0.0, hi=1.0).cpu().numpy(),
                                np.clip(g[..., 0], 0, 1))
  proj = rng.integers(-1, 256, (1, 64, 128, 3)).astype(F32)
  pr = np.clip((proj / F32(255)).astype(F32), 0, 1)
  got = q(torch.from_numpy(proj).to(DEV), torch.float32, div=255.0, lo=0.0, hi=1.0)
  np.testing.assert_array_equal(got.cpu().numpy(), pr)
  np.testing.assert_array_equal(q(got, torch.uint8, mul=255.0, lo=0, hi=255).cpu().numpy(),
                                np.trunc(pr * F32(255)).astype(np.uint8))
  rgb = rng.integers(0, 256, (1, 64, 128, 3)).astype(np.int32)
  np.testing.assert_array_equal(
      q(torch.from_numpy(rgb).to(DEV), torch.float32, div=255.0, lo=-1.0, hi=1.0).cpu().numpy(),
      (rgb / 255).astype(F32))   # int / int: float64 true division, then cast (tf.cast(x / 255))
  u8 = rgb.astype(np.uint8)
  np.testing.assert_array_equal(q(torch.from_numpy(u8).to(DEV), torch.int32, lo=0, hi=255).cpu().numpy(),
                                rgb)
  # edge cases of tf.clip_by_value / tf.cast: a NaN prediction stays NaN through a float clamp
  # (a diverged roll-out must be visible), and the pure int32 -> uint8 cast (lo > hi) wraps
  # modulo 256 (a -1 void class becomes 255), as NumPy's astype does
  nan_in = torch.tensor([float('nan'), -0.5, 0.25, 1.5, float('inf'), -float('inf')], device=DEV)
  got = q(nan_in, torch.float32, lo=0.0, hi=1.0).cpu().numpy()
  assert np.isnan(got[0]) and got[1:].tolist() == [0.0, 0.25, 1.0, 1.0, 0.0]
  sem = np.array([-1, 0, 41, 255, 256, 300, -2], np.int32)
  np.testing.assert_array_equal(q(torch.from_numpy(sem).to(DEV), torch.uint8, lo=1, hi=0).cpu().numpy(),
                                sem.astype(np.uint8))


@pytest.mark.parametrize('n,void', [(1, -1), (2, 0)])
def test_autoregressive_rollout_vs_oracle(n, void):
  """the evaluation roll-out (eval_metric.py:144-239 / gan_manager.py:458-541) as one
  on-device pipeline vs the NumPy / torch oracle, 4 frames.  The oracle is teacher-forced with the
  HIP generator outputs, so the warp / mask / quantisation half of EVERY frame (projected RGB,
  depth, mask, the int32 memory) is compared bit for bit; the generator outputs within 1e-3."""
  from se3ds_amd.utils import eval_metric
  gin_lite.clear_config()
  size, t = 64, 4
  G = image_models.ResNetGenerator(image_size=size, gen_dims=8, z_dim=4, resnet_version='50',
                                   device=DEV, seed=3, dtype=torch.float32)
  _randomise_inference_state(G)
  rng = np.random.default_rng(20 + n)
  image = rng.uniform(0, 1, (n, t, size, 2 * size, 3)).astype(F32)
  depth = rng.uniform(0, 1, (n, t, size, 2 * size, 1)).astype(F32)
  poison = rng.uniform(0, 1, depth.shape)
  depth[poison < 0.02] = 0
  depth[poison > 0.99] = 1
  inputs = dict(image=image, depth=depth, position=(rng.standard_normal((n, t, 3)) * 0.5).astype(F32),
                depth_scale=np.full((n,), 20.0, F32))
  dinputs = {k: torch.from_numpy(v).to(DEV) for k, v in inputs.items()}
  res = eval_metric.generated_rollout(G, dinputs, t, predict_depth=True, unproject_void_class=void)
  feedback = [(g.cpu().numpy(), d.cpu().numpy()) for g, d in zip(res.generated, res.pred_depth)]
  # (frame 0 feeds the ground truth back; its `pred_depth` entry is the target depth)
  ref = model_np.generated_rollout(_cpu_params(G), dict(gen_dims=8, resnet_version='50',
                                                         context_layer='convs', z_dim=4),
                                   inputs, t, True, void, feedback=feedback)
  for k in range(t):
    np.testing.assert_array_equal(res.projected[k].cpu().numpy(), ref['projected'][k], f'frame {k}')
    np.testing.assert_array_equal(res.proj_mask[k].cpu().numpy(), ref['proj_mask'][k], f'frame {k}')
    np.testing.assert_array_equal(res.proj_depth[k].cpu().numpy(), ref['proj_depth'][k], f'frame {k}')
    assert rel_err(res.generated[k].cpu().numpy(), ref['generated'][k]) < 1e-3, k
    if k > 0:
      assert rel_err(res.pred_depth[k].cpu().numpy(), ref['depth_out'][k]) < 1e-3, k
    np.testing.assert_allclose(res.depth_rmse[k].cpu().numpy(), ref['depth_rmse'][k], rtol=1e-5, atol=1e-7)
  assert float(res.proj_mask[0].max()) == 0.0 and float(res.proj_mask[1].mean()) > 0.3
  np.testing.assert_array_equal(res.memory.coords.cpu().numpy(), ref['memory_coords'])
  np.testing.assert_array_equal(res.memory.feats.cpu().numpy(), ref['memory_feats'])
  assert res.memory.m == t * size * 2 * size and res.memory.capacity == res.memory.m
  with pytest.raises(ValueError):
    eval_metric.generated_rollout(G, dinputs, t + 1)
