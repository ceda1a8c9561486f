#!/bin/bash
# build a variant of the library into ab/<name>.so: tools/build_variant.sh name "-DFLAG ..." [file-that-gets-the-flags|all]
set -e
name=$1; flags=$2; only=${3:-all}
rm -rf /tmp/stb_$name && mkdir -p /tmp/stb_$name ab && cp hoigen_amd/csrc/*.hip hoigen_amd/csrc/*.h hoigen_amd/csrc/*.inc /tmp/stb_$name/
cd /tmp/stb_$name && sed -i 's#"../../include/hoigen_amd.h"#"/root/repo/include/hoigen_amd.h"#' hg_api.hip
for f in hg_gemm hg_gemm_ring hg_gemm_ring2 hg_mlp_pair hg_gemm_duo hg_attn hg_qkv_attn hg_vae_fused hg_elem hg_adapter hg_preproc hg_api; do
  fl=""; if [ "$only" = all ] || [ "$only" = $f ]; then fl="$flags"; fi
  ( /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -ffp-contract=fast $fl -c $f.hip -o $f.o 2>&1 | grep -E "error" || true ) &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o /root/repo/ab/$name.so && echo "built ab/$name.so"
