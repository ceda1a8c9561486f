// K loop of the sequence-tile GEMMs for gfx950 (hg_qkv_attn.hip: in_proj + attention; hg_gemm_seq.hip: residual GEMMs):
//
//   acc[208 rows of ONE sequence][384 columns of ONE weight panel] += A[rows][K] W[panel][K]^T
//
// 8 waves, all along N: wave w owns column blocks 3w .. 3w+2 of every row: 13 x 3 accumulator blocks of 16 x 16 = 156 VGPRs.
//   A (fp16 activations, row-major): shared ring of three K-tile stages (208 rows x 128 B, XOR-swizzled 16-byte chunks as in
//     hg_gemm_ring.hip), buffer_load ... lds; every wave reads every row.  Stage 0 sits beside the W rings; stages 1 and 2 share
//     an 80 KiB region with whatever the caller does between two K loops (the attention operands of hg_qkv_attn.hip), so
//     only the FIRST K-tile of the next item is fetched ahead across that phase.
//   W: packed once at load time into MFMA-fragment order (Wp[panel][k / 32][wave][c][lane][8]), streamed from L2 into
//     WAVE-PRIVATE rings of one K-tile (2 x 3 fragments of 1 KiB, read with ds_read_b128 at lane * 16: no swizzle, no sharing,
//     no barrier); a slot is refilled with the next K-tile's fragment as soon as its fragment is in registers.
//   One K-tile = two k-steps of 32 = 26 positions (k-step, row block) of 3 MFMAs each.  The wave's 3 W fragments of a k-step
//   sit in registers; the 13 activation fragments stream through a ring of six registers, five reads ahead of the MFMAs,
//   and the ring runs on ACROSS the K-tile boundary: no barrier and no drained wave there.  One s_barrier per K-tile at
//   position 21, counted vmcnt throughout (below).  Bytes through the CU's load path per K-tile: 26 KiB of A + 48 KiB of W for
//   2 x 13 x 24 MFMAs = 7.4 KB per MFLOP (the 256 x 256 ring: 7.6, the 128 x 256 ring2: 11.4).
#pragma once
#include <type_traits>

#include "hg_gemm_dev.h"

namespace hg {

constexpr int SQ_RB = 13;                          // 16-row blocks of a sequence tile (L <= 208)
constexpr int SQ_NCB = 3;                          // 16-column blocks per wave: 8 waves x 48 = 384 columns per panel
constexpr int SQ_NA = 3;                           // A pieces (1 KiB = 8 rows) per wave and K-tile; waves 0 and 1 issue one more
constexpr int SQ_WSLOT = 2 * SQ_NCB * 1024;        // one wave's W ring: a K-tile of fragments
constexpr int SQ_ASTG = SQ_RB * 2048;              // one A stage: 208 rows x 128 B
constexpr int SQ_A0 = 8 * SQ_WSLOT;                // stage 0
constexpr int SQ_S12 = SQ_A0 + SQ_ASTG;            // stages 1 and 2 (shared with the caller's phase between two K loops)
constexpr int SQ_S12_BYTES = 80 * 1024;
constexpr int SQ_END = SQ_S12 + SQ_S12_BYTES;      // first free byte behind the K loop's LDS
#ifndef SQ_AR
#define SQ_AR 6                                    // activation-fragment ring: SQ_AR - 1 reads ahead (3 or 6: 26 positions = 2 mod SQ_AR)
#endif

#ifndef SQ_PRIO
#define SQ_PRIO 0                                  // s_setprio of the MFMA stream: 0 both waves of a SIMD raised, 1 none, 2 only the younger wave
#endif

// Diagnostic hooks of the loop (s_memtime brackets around its waits): defined by a kernel's HG_STAMPS build, otherwise nothing
#ifndef SQ_STAMP_B
#define SQ_STAMP_B() do {} while (0)
#define SQ_STAMP_E(k) do {} while (0)
#endif

// The loop itself is hg_seq_kloop.inc: a block of lambdas expanded inside the kernel that uses it (a struct of helpers costs the
// compiler its view of the one extern __shared__ array: it then waits vmcnt(0) in front of every ds_read while an LDS-DMA is in
// flight, and the accumulators stop being updated in place).

}  // namespace hg
