# round 3: hardware bf16 conversion -- parity subsets + A/B against the previous library
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
line() { tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', round(d['value'],3), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['conv_ms_per_step'],2), {k:round(v['tflops']) for k,v in d['roofline']['by_kind'].items()})
"; }
timeout 300 python -m pytest tests/test_nets_gpu.py -m gpu -x -q -k "bf16_conversion" 2>&1 | tail -12 | cut -c1-250
echo "== conv_bench new"; N=8 python tools/conv_bench.py 2>&1 | grep -v Warn | grep -v amdgpu.ids
cp se3ds_amd/csrc/libse3ds_hip.so /tmp/new.so
for rep in 1 2; do
  cp se3ds_amd/csrc/_exp/lib_base.so se3ds_amd/csrc/libse3ds_hip.so
  python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "old"
  cp /tmp/new.so se3ds_amd/csrc/libse3ds_hip.so
  python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "new"
done
python bench.py --workload warp --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
SECONDS=0
timeout 1500 python -m pytest tests/test_prod_shapes_gpu.py tests/test_blocks_gpu.py tests/test_nets_gpu.py tests/test_golden_kernels.py tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -5 | cut -c1-250
echo "subset elapsed $SECONDS s"
