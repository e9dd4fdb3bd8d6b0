"""Writes tests/golden/reference_literals.npz: the literal known-answer arrays held by the
reference's OWN unit tests (data, typed in from the assertions; no reference code is copied
or executed -- TensorFlow is not installable here):

  pixel_rays_3      utils/pano_utils_test.py:39-64   equirectangular_pixel_rays(3), (3,6,3)
  pad_input         models/layers_test.py:140-143    4x4 PadLayer input
  pad_const_circ    models/layers_test.py:147-156    PadLayer(2, circular, CONSTANT)
  pad_const_nocirc  models/layers_test.py:158-167    PadLayer(2, non-circular, CONSTANT)
  pad_symm_circ     models/layers_test.py:169-178    PadLayer(2, circular, SYMMETRIC)

Run:  python tests/golden/make_reference_literals.py
"""
import os

import numpy as np

pixel_rays_3 = np.array([
    [[0.0, -1.0, 0.0]] * 6,
    [[0.0, 0.0, -1.0],
     [-9.5105648e-01, 4.3711388e-08, -3.0901703e-01],
     [-5.8778524e-01, 4.3711388e-08, 8.0901694e-01],
     [5.8778524e-01, 4.3711388e-08, 8.0901694e-01],
     [9.5105648e-01, 4.3711388e-08, -3.0901703e-01],
     [0.0, 0.0, -1.0]],
    [[0.0, 1.0, 0.0]] * 6,
], dtype=np.float32)

pad_input = np.array([[1, 3, 2, 2], [1, 1, 2, 2], [1, 1, 2, 2], [2, 0, 3, 3]], np.float32)
pad_const_circ = np.array([
    [0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0],
    [2, 2, 1, 3, 2, 2, 1, 3], [2, 2, 1, 1, 2, 2, 1, 1],
    [2, 2, 1, 1, 2, 2, 1, 1], [3, 3, 2, 0, 3, 3, 2, 0],
    [0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0]], np.float32)
pad_const_nocirc = np.array([
    [0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0],
    [0, 0, 1, 3, 2, 2, 0, 0], [0, 0, 1, 1, 2, 2, 0, 0],
    [0, 0, 1, 1, 2, 2, 0, 0], [0, 0, 2, 0, 3, 3, 0, 0],
    [0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0]], np.float32)
pad_symm_circ = np.array([
    [2, 2, 1, 1, 2, 2, 1, 1], [2, 2, 1, 3, 2, 2, 1, 3],
    [2, 2, 1, 3, 2, 2, 1, 3], [2, 2, 1, 1, 2, 2, 1, 1],
    [2, 2, 1, 1, 2, 2, 1, 1], [3, 3, 2, 0, 3, 3, 2, 0],
    [3, 3, 2, 0, 3, 3, 2, 0], [2, 2, 1, 1, 2, 2, 1, 1]], np.float32)

if __name__ == '__main__':
  out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_literals.npz')
  np.savez(out, pixel_rays_3=pixel_rays_3, pad_input=pad_input, pad_const_circ=pad_const_circ,
           pad_const_nocirc=pad_const_nocirc, pad_symm_circ=pad_symm_circ)
  print('wrote', out)
