"""Per-step loss values and a checksum of the generator / discriminator parameters for the
combinations of the one-replica scheduling switches (dual-stream decoders, per-module optimiser on
the side stream): all combinations must agree BIT FOR BIT (they only reorder independent work).
usage: python tools/step_compare.py [image_size] [batch] [steps]
(SE3DS_CMP_GIN='binding;binding' shrinks the model, tests/test_nets_gpu.py uses it)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == '--worker':
  sys.path.insert(0, ROOT)
  import argparse
  import torch
  from se3ds_amd import bench_step
  size, batch, steps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
  args = argparse.Namespace(image_size=size, dtype='bf16', batch=batch,
                            gin_bindings=[b for b in os.environ.get('SE3DS_CMP_GIN', '').split(';') if b])
  dev = torch.device('cuda', 0)
  gan = bench_step.build_gan(args, dev, 1)
  b = bench_step.synth_batch(batch, size, 1234, dev)
  for step in range(steps):
    gan._reset_metrics()
    gan.train_g_d(b)
    gan.global_step += gan.num_batched_steps
    m = gan._save_metrics_to_dict()
    torch.cuda.synchronize()
    cs = [float(t.double().sum()) for t in (gan.generator.store.theta, gan.discriminator.store.theta,
                                            gan.ema_generator.store.theta, gan.generator.store.state)]
    print('STEP', step, ' '.join(f'{float(m[k]):.9g}' for k in ('dis/disc_loss', 'gen/gen_gan_loss',
                                                                 'gen/depth_loss', 'gen/wc_loss')),
          ' '.join(f'{c:.12e}' for c in cs), flush=True)
  sys.exit(0)

size, batch, steps = (sys.argv[1:] + ['128', '2', '4'])[:3]
# (dual stream, segment optimiser, phases | 'nowg' = WITH the opt-in wgrad stream | 'dov' = WITH the
# opt-in discriminator overlap SE3DS_D_OVERLAP, passes): passes
# 'old' = the round-3 kernels' schedule -- clip and Adam as separate passes over the gradient arena,
# every weight gradient reduced right behind its kernel; '' = round 4 (clip inside Adam, a module's
# split reductions in one deferred launch).  The reference row is the serial order with 'old'.
configs = [('0', '0', '', 'old'), ('0', '0', '', ''), ('0', '1', '', ''), ('1', '0', '', ''), ('1', '1', '', ''),
           ('1', '1', '', 'old'), ('1', '0', 'fwd', ''), ('1', '0', 'bwd', ''), ('1', '1', 'nowg', ''),
           ('1', '0', 'nowg', 'old'), ('1', '1', 'dov', '')]
if os.environ.get('SE3DS_CMP_CONFIGS'):   # e.g. '0:0::old,1:1::' -- a subset (the reference row must be in it)
  configs = [tuple((c.split(':') + [''])[:4]) for c in os.environ['SE3DS_CMP_CONFIGS'].split(',')]
out = {}
for ds, so, ph, ps in configs:
  env = dict(os.environ, SE3DS_DUAL_STREAM=ds, SE3DS_SEGMENT_OPTIMIZER=so,
             SE3DS_FUSED_CLIP_ADAM='0' if ps == 'old' else '1',
             SE3DS_DEFER_WGRAD_REDUCE='0' if ps == 'old' else '1',
             SE3DS_DUAL_PHASES='' if ph in ('nowg', 'dov') else ph,
             SE3DS_WGRAD_STREAM='1' if ph == 'nowg' else '0', SE3DS_D_OVERLAP='1' if ph == 'dov' else '0')
  r = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', size, batch, steps],
                     env=env, capture_output=True, text=True)
  lines = [l for l in r.stdout.splitlines() if l.startswith('STEP')]
  out[(ds, so, ph, ps)] = lines
  print(f'--- dual_stream={ds} segment_optimizer={so} phases={ph or "fwd,bwd"} passes={ps or "round4"} rc={r.returncode}')
  print('\n'.join(lines) if lines else r.stderr[-2000:])
ref = out[('0', '0', '', 'old')]
bad = 0
for k, v in out.items():
  first = next((i for i, (a, b) in enumerate(zip(v, ref)) if a != b), None)
  same = v == ref and len(v) == int(steps)
  bad += not same
  print(k, 'IDENTICAL to serial' if same else f'DIFFERS from serial, first at step {first}')
sys.exit(1 if bad else 0)
