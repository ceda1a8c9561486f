#!/bin/bash
# one-kernel VAE: product build, stamps, knock-outs (compile-time -DVF_XMODE) and read-ahead depths (ab/vf*.so variants of hg_vae_fused.hip)
ROOT=$GRAFT_REPO_ROOT; mkdir -p $ROOT/gpurun_out; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_vae_fused.py -m gpu -x -q 2>&1 | tail -3
export HG_VAE_FUSED=2 R=98304 ITERS=5
run() { echo "== $*"; env "$@" timeout 300 python tools/vae_time.py 2>&1 | grep -v amdgpu.ids | tail -3; }
run A=product
run HG_LIB_PATH=$ROOT/ab/vfs.so HG_VF_STAMPS=1
for v in vfx1 vfx4 vfa3 vfa6; do run HG_LIB_PATH=$ROOT/ab/$v.so; done
