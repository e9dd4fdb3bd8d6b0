# one measured attempt at co-running the HBM-bound kernels under the convolutions.
# The decoders' two streams (and / or the optimiser's) get CU masks (SE3DS_CU_MASK, hipops/nn.py
# make_stream): a kernel of a masked stream runs on ITS CUs whatever the other stream's persistent conv
# workgroups hold.  A/B of the default schedule inside one call, 2 reps.
#   gpurun -- 'bash tools/probes/cu_mask_ab.sh'
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/gpu.sh ab SE3DS_CU_MASK=halves SE3DS_CU_MASK=alternate SE3DS_CU_MASK=opt32 SE3DS_CU_MASK=opt64
