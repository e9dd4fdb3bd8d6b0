"""What a library GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on the 1x1-conv shapes (bf16),
as a yardstick for igemm_big / igemm_glds / wgrad_glds on the same shapes."""
import torch
DEV = 'cuda:0'
shapes = [(16384, 2048, 512), (16384, 512, 2048), (65536, 1024, 256), (65536, 256, 1024), (16384, 1024, 4096 // 4)]
for M, K, N in shapes:
  a = torch.randn((M, K), device=DEV).bfloat16()
  b = torch.randn((K, N), device=DEV).bfloat16()
  for _ in range(5):
    c = a @ b
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(50):
    c = a @ b
  e1.record()
  torch.cuda.synchronize()
  us = e0.elapsed_time(e1) * 1e3 / 50
  print(f'fwd-like  [{M}x{K}]x[{K}x{N}]: {us:.1f} us  {2.0 * M * K * N / us * 1e-6:.0f} TFLOP/s')
  at = a.t().contiguous()   # wgrad-like: [K x M] x [M x N]
  g = torch.randn((M, N), device=DEV).bfloat16()
  for _ in range(5):
    w = a.t() @ g
  torch.cuda.synchronize()
  e0.record()
  for _ in range(50):
    w = a.t() @ g
  e1.record()
  torch.cuda.synchronize()
  us = e0.elapsed_time(e1) * 1e3 / 50
  print(f'wgrad-like [{K}x{M}]x[{M}x{N}]: {us:.1f} us  {2.0 * M * K * N / us * 1e-6:.0f} TFLOP/s')
