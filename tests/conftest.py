import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


# Host threads of the CPU oracle.  Measured on the GPU boxes (2 x 64-core EPYC, 256 hardware
# threads; tools/oracle_f64_time.py): the oracle's generator forward + backward at the yardstick
# test's size takes 75 s (fp32) / 25 s (fp64) on torch's default of 128 threads and 12 s / 5 s on
# 16 -- the work is dominated by weight-sized passes and small-M matrix products, which the
# default oversubscribes across two sockets.  8 ... 24 threads are within 15 % of each other.
ORACLE_THREADS = 16


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box)')
  import torch
  torch.set_num_threads(max(1, min(ORACLE_THREADS, os.cpu_count() or 1)))


@pytest.fixture(scope='session')
def golden_dir():
  return os.path.join(ROOT, 'tests', 'golden')
