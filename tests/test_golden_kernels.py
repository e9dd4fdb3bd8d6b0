"""Golden kernel fixtures (tests/golden/kernel_fixtures.npz, made by make_kernel_fixtures.py):
CPU: the oracle reproduces the stored outputs from the stored inputs; GPU: the HIP path (through
the C ABI) matches the stored outputs without running the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import nets_torch as O
from se3ds_amd.hipops import nn
from tests import test_prod_shapes_gpu as P
from tests.golden import make_kernel_fixtures as M

FIX = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'kernel_fixtures.npz'))
DEV = 'cuda:0'


def _conv_inputs(i):
  g = lambda k: torch.from_numpy(FIX[f'conv{i}/in/{k}']) if f'conv{i}/in/{k}' in FIX else None
  return g('x'), g('kern'), g('b'), g('u'), g('mask'), g('gy')


def _conv_outputs(i):
  return {k: (FIX[f'conv{i}/out/{k}'] if f'conv{i}/out/{k}' in FIX else None)
          for k in ('y', 'dx', 'dk', 'db', 'um')}


@pytest.mark.parametrize('i', range(len(M.CONV_CASES)))
def test_oracle_reproduces_conv_fixture(i):
  case = M.CONV_CASES[i]
  ref = P._oracle(case, case[12][0], inputs=_conv_inputs(i))
  want = _conv_outputs(i)
  for k, v in want.items():
    if v is not None:
      assert P.rel_err(ref[k], v) < 1e-5, (case[0], k)


@pytest.mark.parametrize('i', range(len(M.NORM_CASES)))
def test_oracle_reproduces_norm_fixture(i):
  case = M.NORM_CASES[i]
  d = {k: torch.from_numpy(FIX[f'norm{i}/{k}']) for k in ('gamma', 'beta', 'x', 'r', 'gy')}
  out = M.norm_oracle(case, d)
  for k, v in out.items():
    assert P.rel_err(v.detach().numpy(), FIX[f'norm{i}/{k}']) < 1e-5, (case, k)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('i', range(len(M.CONV_CASES)))
def test_hip_conv_matches_fixture(i, dtype):
  case = M.CONV_CASES[i]
  got = P._hip(case, case[12][0], dtype, inputs=_conv_inputs(i))
  want = _conv_outputs(i)
  if dtype == torch.float32:
    t_act = t_par = P.TOL_F32
  else:
    t_act = P.TOL_BF16_STORED
    t_par = P.TOL_BF16_F32OUT_SCALED_DY if case[9] else P.TOL_BF16_F32OUT
  if want['um'] is not None:
    np.testing.assert_array_equal(got['um'], want['um'])
  for k, t in (('y', t_act), ('dx', t_act), ('dk', t_par), ('db', t_par)):
    if want[k] is not None:
      assert P.rel_err(got[k], want[k]) < t, (case[0], str(dtype), k, P.rel_err(got[k], want[k]))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('i', range(len(M.NORM_CASES)))
def test_hip_norm_matches_fixture(i, dtype):
  kind, n, h, w, c, act, with_res = M.NORM_CASES[i]
  f = lambda k: torch.from_numpy(FIX[f'norm{i}/{k}'])
  store = nn.ParamStore()
  layer = nn.NormLayer(store, 'n', c, kind)
  store.finalize(DEV, None)
  store.load_dict({'n/gamma': FIX[f'norm{i}/gamma'], 'n/beta': FIX[f'norm{i}/beta']})
  ctx = nn.Ctx(DEV, dtype, training=True, record=True)
  xv = nn.Var(f('x').to(DEV).to(dtype))
  rv = nn.Var(f('r').to(DEV).to(dtype)) if with_res else None
  yv = nn.norm_act(ctx, xv, layer, act=act, alpha=M.ALPHA, res=rv)
  yv.grad = f('gy').to(DEV).to(dtype)
  ctx.backward()
  t_act = P.TOL_F32 if dtype == torch.float32 else 2 * P.TOL_BF16_STORED
  t_par = 2e-4 if dtype == torch.float32 else 1e-2
  assert P.rel_err(yv.data.float().cpu().numpy(), FIX[f'norm{i}/y']) < t_act
  assert P.rel_err(xv.grad.float().cpu().numpy(), FIX[f'norm{i}/dx']) < 2 * t_act
  assert P.rel_err(store.grad_views['n/gamma'].cpu().numpy(), FIX[f'norm{i}/dgamma']) < t_par
  assert P.rel_err(store.grad_views['n/beta'].cpu().numpy(), FIX[f'norm{i}/dbeta']) < t_par
  if with_res:
    assert P.rel_err(rv.grad.float().cpu().numpy(), FIX[f'norm{i}/dres']) < t_act
  if kind == 'batch':
    for nm in ('moving_mean', 'moving_variance'):
      assert P.rel_err(store['n/' + nm].cpu().numpy(), FIX[f'norm{i}/{nm}']) < 1e-5, nm
