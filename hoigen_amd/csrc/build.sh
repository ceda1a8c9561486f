#!/bin/bash
# Build libhoigen_amd.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function -ffp-contract=fast ${HG_EXTRA_FLAGS}"   # HG_EXTRA_FLAGS=-DHG_STAMPS: diagnostic build with s_memtime stamps
OBJS=""
for f in hg_gemm hg_gemm_ring hg_gemm_ring2 hg_mlp_pair hg_gemm_duo hg_attn hg_qkv_attn hg_vae_fused hg_elem hg_adapter hg_preproc hg_api; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ hg_kernels.h -nt $f.o ] || [ hg_common.h -nt $f.o ] || [ hg_gemm_dev.h -nt $f.o ] || [ hg_gemm_ring_body.h -nt $f.o ] || [ hg_gemm_ring2_body.h -nt $f.o ] || [ hg_attn_dev.h -nt $f.o ] || [ hg_seq_dev.h -nt $f.o ] || [ hg_seq_kloop.inc -nt $f.o ] || [ hg_seq_kloop_run.inc -nt $f.o ] || [ hg_qkv_attn_body.inc -nt $f.o ] || [ ../../include/hoigen_amd.h -nt $f.o ]; then
    rm -f $f.o
    ( $HIPCC $FLAGS -c $f.hip -o $f.o || rm -f $f.o ) &
  fi
  OBJS="$OBJS $f.o"
done
wait
for o in $OBJS; do [ -f $o ] || { echo "compile failed: $o"; exit 1; }; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined $OBJS -o libhoigen_amd.so
echo "built $(pwd)/libhoigen_amd.so"
