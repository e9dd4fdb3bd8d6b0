"""Builds libse3ds_hip.so (gfx950) in-tree with hipcc: one object per .hip file, compiled in
parallel, linked into se3ds_amd/csrc/libse3ds_hip.so.  Cross-compiles without a GPU.

  python -m se3ds_amd.csrc.build [--force]
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'libse3ds_hip.so')
OBJ_DIR = os.path.join(HERE, '_obj')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
          '-fno-fast-math']
# Geometry/index math must round every fp32 op separately (bit-exact indices);
# the contraction kernels are free to fuse.
PER_FILE = {
    'geom.hip': ['-ffp-contract=off'],
    'input.hip': ['-ffp-contract=off'],
}


def sources():
  return sorted(f for f in os.listdir(HERE) if f.endswith('.hip'))


def _deps():
  deps = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.h')]
  inc = os.path.join(HERE, '..', '..', 'include')
  deps += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith('.h')]
  deps.append(os.path.abspath(__file__))
  return deps


def _stale(target, srcs):
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(s) > t for s in srcs)


def _compile(src):
  obj = os.path.join(OBJ_DIR, src.replace('.hip', '.o'))
  path = os.path.join(HERE, src)
  if not _stale(obj, [path] + _deps()):
    return obj, None
  cmd = [HIPCC] + COMMON + PER_FILE.get(src, []) + ['-c', path, '-o', obj]
  r = subprocess.run(cmd, capture_output=True, text=True)
  if r.returncode != 0:
    raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
  return obj, r.stderr


def build(force=False, verbose=False):
  os.makedirs(OBJ_DIR, exist_ok=True)
  srcs = sources()
  if force:
    for f in os.listdir(OBJ_DIR):
      os.remove(os.path.join(OBJ_DIR, f))
  with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
    results = list(ex.map(_compile, srcs))
  objs = [o for o, _ in results]
  if verbose:
    for _, err in results:
      if err:
        sys.stderr.write(err)
  if force or _stale(OUT, objs):
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
  return OUT


if __name__ == '__main__':
  print(build(force='--force' in sys.argv, verbose=True))
