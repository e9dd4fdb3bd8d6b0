# round 3: 128-channel halo kernel with one 16-MFMA slot pair per K step (library variant), and the
# 256-pixel macro tile vs the 128 x 128 tile on the 1x1 shapes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
line() { tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', round(d['value'],3), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['conv_ms_per_step'],2), {k:round(v['tflops']) for k,v in d['roofline']['by_kind'].items()})
"; }
echo "== conv_bench default"; N=8 python tools/conv_bench.py 2>&1 | grep -v Warn
echo "== conv_bench BIG_TILE=0"; N=8 SE3DS_BIG_TILE=0 python tools/conv_bench.py 2>&1 | grep "1x1"
python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "default"
SE3DS_BIG_TILE=0 python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "big_tile=0"
cp se3ds_amd/csrc/libse3ds_hip.so /tmp/base.so
cp se3ds_amd/csrc/libse3ds_hip_qp4.so se3ds_amd/csrc/libse3ds_hip.so
echo "== conv_bench qp4"; N=8 python tools/conv_bench.py 2>&1 | grep "128->128"
timeout 900 python -m pytest tests/test_prod_shapes_gpu.py tests/test_blocks_gpu.py -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "qp4"
cp /tmp/base.so se3ds_amd/csrc/libse3ds_hip.so
python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "default"
cp se3ds_amd/csrc/libse3ds_hip_qp4.so se3ds_amd/csrc/libse3ds_hip.so
python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "qp4"
cp /tmp/base.so se3ds_amd/csrc/libse3ds_hip.so
