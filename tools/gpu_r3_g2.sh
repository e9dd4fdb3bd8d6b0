cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
SECONDS=0
timeout 1200 python -m pytest tests/test_blocks_gpu.py -m gpu -q -s -k "patch_discriminator" 2>&1 | grep -v "^$" | tail -30 | cut -c1-300
echo "elapsed $SECONDS s"
