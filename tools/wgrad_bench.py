"""Steady-state timing of se3ds_conv2d_wgrad alone (bf16, stride 1): ITERS back-to-back launches
between two events, after a warm-up.   python tools/wgrad_bench.py [cin cout k h w n]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from se3ds_amd import _lib   # noqa: E402
from se3ds_amd.hipops import nn   # noqa: E402,F401  (registers the signatures)
L = _lib.lib()

DEV = 'cuda:0'
cin, cout, k, h, w, n = (int(a) for a in (sys.argv[1:7] if len(sys.argv) >= 7 else (1024, 1024, 3, 32, 64, 8)))
pad = k // 2
x = torch.randn((n, h, w, cin), device=DEV).bfloat16()
dy = torch.randn((n, h, w, cout), device=DEV).bfloat16()
dw = torch.zeros((k, k, cin, cout), device=DEV)
wsz = L.se3ds_conv2d_wgrad_workspace_bytes(n, h, w, cin, cout, k, k)
ws = torch.empty(wsz, dtype=torch.uint8, device=DEV)
stream = torch.cuda.current_stream().cuda_stream
mask = None
if os.environ.get('MASK'):
  mask = (torch.rand((n, h, w), device=DEV) > 0.3).float()
def run():
  rc = L.se3ds_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 3, n, h, w, cin, h, w, cout, k, k, 1,
                            pad, pad, 0, mask.data_ptr() if mask is not None else None, 1 if mask is not None else 0,
                            None, None, 0, ws.data_ptr(), wsz, stream)
  assert rc == 0, rc
for _ in range(10):
  run()
torch.cuda.synchronize()
iters = int(os.environ.get('ITERS', '50'))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
  run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / iters
fl = 2.0 * k * k * cin * cout * n * h * w
print(f'wgrad {cin}->{cout} k{k} @{h}x{w} n{n}: {us:.1f} us  {fl / us * 1e-6:.0f} TFLOP/s  mask={int(mask is not None)}')
