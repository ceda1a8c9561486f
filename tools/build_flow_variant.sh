#!/bin/bash
# ab/flow.so: the product library plus tools/experiments/round3/hg_gemm_flow.hip (test-hook kernel id 4; HG_FLOW=1 routes the
# residual GEMMs of the step to it) - for A/B runs only
set -e
rm -rf /tmp/stb_flow && mkdir -p /tmp/stb_flow && cp hoigen_amd/csrc/*.hip hoigen_amd/csrc/*.h /tmp/stb_flow/ && cp tools/experiments/round3/hg_gemm_flow.hip /tmp/stb_flow/
cd /tmp/stb_flow && python3 - <<'PY'
s=open("hg_api.hip").read().replace('"../../include/hoigen_amd.h"','"/root/repo/include/hoigen_amd.h"')
for tag in ("    ps.finish();", '    if (e != hipSuccess) return fail(c, HG_ERR_HIP, "test gemm (ln) launch failed'):
    s=s.replace("    else e = launch_gemm(epi, g, s);\n"+tag, "    else if (kernel == 4) e = gemm_flow_ok(epi, g) ? launch_gemm_flow(epi, g, s) : hipErrorInvalidValue;\n    else e = launch_gemm(epi, g, s);\n"+tag)
open("hg_api.hip","w").write(s)
s=open("hg_kernels.h").read()
s=s.replace("hipError_t launch_gemm_ring2(int epi, const GemmArgs& a, hipStream_t s);","hipError_t launch_gemm_ring2(int epi, const GemmArgs& a, hipStream_t s);\nbool gemm_flow_ok(int epi, const GemmArgs& a);\nhipError_t launch_gemm_flow(int epi, const GemmArgs& a, hipStream_t s);")
open("hg_kernels.h","w").write(s)
s=open("hg_gemm.hip").read()
s=s.replace("    if (epi == EPI_SCALE_RESID_LN_F32) return gemm_duo_ok(epi, a)","    static const int flow = []() { const char* e = getenv(\"HG_FLOW\"); return e ? atoi(e) : 0; }();\n    if (flow && !force_simple && (epi == EPI_RESID_LN_F32 || epi == EPI_BIAS_RESID_F32) && gemm_flow_ok(epi, a)) return launch_gemm_flow(epi, a, s);\n    if (epi == EPI_SCALE_RESID_LN_F32) return gemm_duo_ok(epi, a)",1)
open("hg_gemm.hip","w").write(s)
for f in ("hg_gemm_flow.hip","hg_gemm_ring2.hip"):
    s=open(f).read()
    s=s.replace("    const int grid = n_tiles < n_cu ? n_tiles : n_cu;","    int grid = n_tiles < n_cu ? n_tiles : n_cu;\n    if (const char* e = getenv(\"HG_RING2_GRID\")) { const int v = atoi(e); if (v >= 8 && v < grid) grid = v; }")
    open(f,"w").write(s)
PY
for f in hg_gemm hg_gemm_ring hg_gemm_ring2 hg_gemm_flow hg_gemm_duo hg_attn hg_elem hg_adapter hg_preproc hg_api; do ( /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -ffp-contract=fast $FLOW_FLAGS -c $f.hip -o $f.o 2>&1 | grep -E "error" || true ) & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o /root/repo/ab/${FLOW_NAME:-flow}.so && echo "built ab/${FLOW_NAME:-flow}.so"
