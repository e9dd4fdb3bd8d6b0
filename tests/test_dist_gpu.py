"""Two replicas on one MI355X (gloo between them): the multi-replica trainer path end to end."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# the GPU boxes' hostname does not resolve: pin the rendezvous transports to the loopback interface
_LOOPBACK = dict(GLOO_SOCKET_IFNAME='lo', NCCL_SOCKET_IFNAME='lo')


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


@pytest.mark.parametrize('own_comm', ['0', '1'])
def test_two_replicas_share_one_gpu_over_gloo(own_comm):
  """Product DP step on 2 replicas vs the R-replica oracle (see the worker).  own_comm=1 also runs
  the opt-in second communicator for the gradient traffic.  A hang is a FAILURE: the workers dump
  their stacks (faulthandler) and exit non-zero after 170 s."""
  env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0',
             SE3DS_GRAD_SYNC_OWN_COMM=own_comm, **_LOOPBACK)
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
         '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
         os.path.join(ROOT, 'tests', '_dist_gpu_worker.py')]
  # ONE attempt.  Round 2 retried here ("a gloo rendezvous on a fresh box hung once in ~10 runs").
  # The box's hostname does not resolve (c10d logs "The hostname of the client socket cannot be
  # retrieved. err=-3" on every run), and without an interface name gloo derives its listening
  # address from exactly that lookup; _LOOPBACK pins both transports to `lo`, which removes the
  # lookup from the rendezvous.  A hang is still visible: the workers dump their stacks after 170 s.
  r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
  assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
  assert 'DIST_GPU_OK' in r.stdout
  print(r.stdout[-600:])


def test_two_replica_stream_schedule_is_bit_identical_to_one_stream():
  """With several replicas the decoders can keep their two HIP streams (lockstep branch threads,
  event-ordered paired SyncBN sums; opt-in SE3DS_DUAL_STREAM_DP=1 until it has run over RCCL on two
  real GPUs -- gloo blocks the host and cannot show a stream-ordering mistake around NCCL's streams)
  and the per-module fix-up / clip / gradient hand-over runs on the optimiser's side stream -- the
  schedule the one-GPU bench measures.  It only reorders independent work: the two-replica worker's signature (parameter / Adam / EMA checksums after
  train_g_d, train_d, train_g_d) must equal the one-stream run's BIT FOR BIT, and the SyncBN
  collective count stays 2 x (all batch norms - those of one decoder branch)."""
  sigs = {}
  for dual in ('0', '1'):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', SE3DS_DUAL_STREAM='1',
               SE3DS_DUAL_STREAM_DP=dual, SE3DS_GRAD_SYNC_OWN_COMM='0', **_LOOPBACK)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'tests', '_dist_gpu_worker.py')]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    line = [l for l in r.stdout.splitlines() if 'DIST_GPU_OK' in l][-1]
    sigs[dual] = line
    assert ('streams=2' in r.stdout) == (dual == '1'), r.stdout[-800:]
  assert sigs['0'] == sigs['1'], sigs


def test_rccl_single_rank_api_paths():
  """The real `nccl` (= RCCL) backend with one rank: every collective shape of the multi-GPU step
  is accepted (the 8-GPU run itself is the driver's)."""
  env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', **_LOOPBACK)
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_rccl_worker.py'),
                      str(_free_port())], env=env, cwd=ROOT, capture_output=True, text=True,
                     timeout=240)
  assert r.returncode == 0 and 'RCCL_OK' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def _run_bench(args, env_extra, timeout=600):
  env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', **_LOOPBACK)
  env.update(env_extra)
  env.pop('WORLD_SIZE', None)
  env.pop('RANK', None)
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, cwd=ROOT,
                     capture_output=True, text=True, timeout=timeout)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  import json
  line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
  return json.loads(line)


def test_bench_gpus_2_launches_two_ranks_on_one_gpu():
  """`bench.py --gpus 2` (no torchrun around it) starts two ranks itself and reports n_gpus = 2.
  On a 1-GPU box the ranks share cuda:0 and rendezvous over gloo (SE3DS_BENCH_BACKEND); the warp
  workload is replicas-only, so this exercises exactly the launcher / barrier / max-over-ranks
  plumbing of the multi-GPU bench."""
  out = _run_bench(['--gpus', '2', '--workload', 'warp', '--warp-height', '64', '--steps', '3',
                    '--warmup', '1', '--no-cpu-baseline'], dict(SE3DS_BENCH_BACKEND='gloo'))
  assert out['n_gpus'] == 2 and out['steps'] == 3 and out['value'] > 0
  assert out['scaling'] == 'weak' and out['config']['workload'].startswith('warp')
  # the pre-flight block of a multi-rank run: both ranks seen, the all-reduce really summed
  # (ones all-reduced 1 + 5 times over 2 ranks = 64), bandwidth / latency figures present
  pre = out['collectives']
  assert pre['rccl_ranks_seen'] == [0, 1] and pre['sum_check'] == 64.0, pre
  assert pre['allreduce_big_busbw_GBs'] > 0 and pre['allreduce_syncbn_2x1024_us'] > 0, pre


def test_bench_gan_step_two_gloo_ranks_on_one_gpu():
  """PLUMBING ONLY: `bench.py --gpus 2` with the gan_step workload, both ranks on the one GPU over
  gloo (small model: the figure says nothing about throughput) -- the launcher, the data-parallel
  strategy with branch streams and side-stream gradient hand-over, barrier + max-over-ranks timing
  and the line's n_gpus / global_batch / backend fields."""
  out = _run_bench(['--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '1', '--image-size',
                    '128', '--no-cpu-baseline', '--no-warp', '--no-shipped', '--no-batch-max'],
                   dict(SE3DS_BENCH_BACKEND='gloo'), timeout=900)
  assert out['n_gpus'] == 2 and out['config']['global_batch'] == 2 and out['backend'] == 'gloo'
  assert all(v == v for v in out['losses'].values()) and out['value'] > 0
  # what one rank sends per step at the REAL model dimensions (ResNet-101, gen_dims 128): the SyncBN
  # sums of the encoder / context one by one, the two decoders' in pairs, and the whole clipped
  # gradient arena (1.114 B generator + 30.5 M discriminator parameters, fp32) in buckets
  cps = out['collectives_per_step']
  print('collectives per step:', cps)
  singles, pairs = cps.get('syncbn', {'count': 0})['count'], cps.get('syncbn_pair', {'count': 0})['count']
  # forward + backward: every one of the generator's 279 batch norms is summed once per pass, the
  # k-th norms of the two decoder branches (84 decoder norms + 1 head norm each) share a collective
  assert singles % 2 == 0 and pairs % 2 == 0 and singles // 2 + 2 * (pairs // 2) == 279, cps
  assert pairs // 2 == 85 and singles + pairs == 2 * (279 - 85), cps
  grad_bytes = sum(cps.get(k, {'bytes': 0})['bytes'] for k in ('grad_bucket', 'grad_arena'))
  assert 4.4e9 < grad_bytes < 4.8e9, cps
  # round 6: the same step seen from the device -- HIP events around every collective on the stream
  # it was issued on -- and every rank's own step time, so that the first scaling file explains itself
  dev_side = cps['device']
  by = dev_side['by_kind']
  assert by['syncbn']['count'] == singles and by['syncbn_pair']['count'] == pairs, by
  assert by['grad_bucket']['count'] == cps['grad_bucket']['count'] and by['finish_wait']['count'] >= 1, by
  assert dev_side['exposed_syncbn_ms'] > 0 and dev_side['grad_bucket_ms'] > 0 and dev_side['finish_wait_ms'] >= 0
  assert dev_side['finish_buckets'] is not None and dev_side['finish_buckets'] >= 0
  assert len(out['ms_per_step_by_rank']) == 2 and max(out['ms_per_step_by_rank']) <= out['ms_per_step'] * 1.001


def test_gan_step_on_two_gpus_over_rccl():
  """Two real GPUs, RCCL: the data-parallel GAN step (SyncBN statistics all-reduces on the main
  stream, gradient buckets on the side stream) runs and reports a whole-job value for 2 ranks.
  Skipped on a 1-GPU box (the 8-GPU scaling run is the driver's)."""
  import torch
  if torch.cuda.device_count() < 2:
    pytest.skip('needs two GPUs')
  for own, dual in (('0', '0'), ('1', '0'), ('0', '1')):
    out = _run_bench(['--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '1', '--image-size',
                      '128', '--no-cpu-baseline', '--no-warp'],
                     dict(SE3DS_GRAD_SYNC_OWN_COMM=own, SE3DS_DUAL_STREAM_DP=dual))
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 2
    assert all(v == v for v in out['losses'].values())
