import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import warp_c, warp_np
from se3ds_amd.utils import pano_utils
from tests.test_warp_gpu import synth_pano, t
rng = np.random.default_rng(12)
for h in (256, 512):
  w = 2*h
  rgb, depth = synth_pano(rng, 1, h, w)
  posn = rng.standard_normal((1, 3)).astype(np.float32)
  pos = t(posn)
  xyz1, f = pano_utils.equirectangular_to_pointcloud(t(rgb), t(depth), -1, 20.0, position=pos)
  pd, prgb, pm = pano_utils.project_feats_to_equirectangular(f, xyz1, h, w, -1, 20.0, offset=pos, with_mask=True)
  d_o, f_o = warp_c.project_feats_to_equirectangular(f.cpu().numpy(), xyz1.cpu().numpy(), h, w, -1, 20.0, offset=posn)
  got = pd.cpu().numpy()
  valid = (depth > 0) & (depth < 1)
  print(h, 'gpu==oracle depth', np.mean(got == d_o), 'feat', np.mean(prgb.cpu().numpy() == f_o))
  rel = np.abs(got[valid] - depth[valid]) / depth[valid]
  print('  frac rel>1e-4 gpu', (rel > 1e-4).mean(), ' oracle', (np.abs(d_o[valid]-depth[valid])/depth[valid] > 1e-4).mean(), 'pos', posn)
  bad = np.argwhere((np.abs(got - depth)/np.maximum(depth,1e-9) > 1e-4) & valid)[:5]
  for b in bad: print('   ', b, got[tuple(b)], depth[tuple(b)], d_o[tuple(b)])
