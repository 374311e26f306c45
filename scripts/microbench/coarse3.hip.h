// coarse3.hip.h -- third-generation sweep of the MFMA pre-filter (round 3).
//
// Same contract and the same visit / flush machinery as coarse2_kernel (coarse2.hip.h); what changes is the
// instruction stream of the sweep.  Measured facts it is built on (profiles/r02_mfma_shadow.txt,
// r02_mfma_valu_prio.txt, r02_pmc_variants.txt): on gfx950 a wave's VALU work overlaps matrix work only when
// it sits in the shadow of the SAME wave's MFMAs (an MFMA-only wave and a VALU-only wave on one SIMD take the
// sum of their times); about five VALU instructions per 32x32x16 MFMA are covered.  coarse2 puts the skip
// test of the first unit of a tile behind the MFMAs of the second and leaves the second unit's test (and the
// LDS wait of every tile) exposed: MFMA pipe 59 % busy with visits disabled, 31 % as shipped.
//
//   * Software pipeline with ONE unit of lag: the KS MFMAs of unit u+1 are issued first, the min tree of unit
//     u runs in their shadow, tile after tile (the pending unit is carried across tiles and stages in its
//     accumulator registers).  No VALU instruction of the sweep is outside an MFMA shadow of its own wave.
//   * |r'|^2 is no longer loaded into the accumulator in place: it lives in a register set of its own (cb) that
//     both units of a tile read as the C operand (D != C), loaded one tile ahead together with the hi
//     fragments -- the LDS latency of a tile is covered by the previous tile's MFMAs.
//   * That costs 24 VGPRs (second cb / hi set): 12 waves per CU (3 per SIMD, <= 168 VGPRs) instead of 16.
#pragma once
#include "../../sknnr_amd/csrc/coarse2.hip.h"

namespace sknnr {

// One unit of the pipelined sweep:
//   X  = sum_s ah[s] . bq[s] + cb          (KS MFMAs; cb holds |r'|^2 in accumulator order, D != C)
//   g, m = min tree of Y                    (the PREVIOUS unit's values, complete by now) in the MFMAs' shadow
// Hazards (guide section 5.7): the first read of Y must come >= 11 wait states after the last MFMA that wrote
// it.  That MFMA is the last one of the previous unit_step; behind it sit, at least, the rest of that step's
// tree (KS <= 2: 5 instructions), the v_cmp and the s_cbranch of the skip test, this step's first MFMA and the
// s_nop padding below -- 12 states or more for every KS.  X is first read by the NEXT step's tree, same rule.
struct NoPrefetch {
    __device__ __forceinline__ void operator()() const {}
};
// `prefetch` runs right behind the first MFMA: the LDS requests for the NEXT tile's operands go out there, so that the
// compiler's (uncounted) wait in front of the next tile's first MFMA finds them a whole tile old.
template <int KS, typename Prefetch = NoPrefetch>
__device__ __forceinline__ void unit_step(floatx16& X, const floatx16& Y, const floatx16& cb, const half8 (&ah)[KS],
                                          const half8 (&bq)[KS], float (&g)[5], float& m, Prefetch prefetch = Prefetch()) {
    static_assert(KS >= 1 && KS <= 4, "hand-scheduled for one to four K-steps");
    if constexpr (KS <= 2) {
        asm volatile("v_mfma_f32_32x32x16_f16 %[X], %[h0], %[q0], %[cb]\n\ts_nop 3\n\t"
                     : [X] "=&v"(X)
                     : [h0] "v"(ah[0]), [q0] "v"(bq[0]), [cb] "v"(cb), "v"(Y));
    } else {
        asm volatile("v_mfma_f32_32x32x16_f16 %[X], %[h0], %[q0], %[cb]\n\ts_nop 7\n\ts_nop 1\n\t"
                     : [X] "=&v"(X)
                     : [h0] "v"(ah[0]), [q0] "v"(bq[0]), [cb] "v"(cb), "v"(Y));
    }
    asm volatile("" ::: "memory");
    prefetch();
    asm volatile("" ::: "memory");
    const floatx16& x = Y;
    asm volatile(SKNNR_TREE_A
                 : [g0] "=&v"(g[0]), [g1] "=&v"(g[1]), [g2] "=&v"(g[2]), [g3] "=&v"(g[3])
                 : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]),
                   [x6] "v"(x[6]), [x7] "v"(x[7]), [x8] "v"(x[8]), [x9] "v"(x[9]), [x10] "v"(x[10]), [x11] "v"(x[11]));
    if constexpr (KS >= 2)
        asm volatile("v_mfma_f32_32x32x16_f16 %[X], %[h1], %[q1], %[X]\n\t" : [X] "+v"(X) : [h1] "v"(ah[1]), [q1] "v"(bq[1]));
    asm volatile("v_min3_f32 %[g4], %[x12], %[x13], %[x14]\n\t"
                 "v_min3_f32 %[m], %[g0], %[g1], %[g2]\n\t"
                 "v_min_f32 %[g4], %[g4], %[x15]\n\t"
                 "s_nop 0\n\t"
                 "v_min3_f32 %[m], %[m], %[g3], %[g4]\n\t"
                 : [g4] "=&v"(g[4]), [m] "=&v"(m)
                 : [x12] "v"(x[12]), [x13] "v"(x[13]), [x14] "v"(x[14]), [x15] "v"(x[15]), [g0] "v"(g[0]), [g1] "v"(g[1]),
                   [g2] "v"(g[2]), [g3] "v"(g[3]));
#pragma unroll
    for (int s = 2; s < KS; ++s)
        asm volatile("v_mfma_f32_32x32x16_f16 %[X], %[h], %[q], %[X]\n\t" : [X] "+v"(X) : [h] "v"(ah[s]), [q] "v"(bq[s]));
}

// The MFMAs of a unit alone (pipeline prologue: nothing is pending yet).
template <int KS>
__device__ __forceinline__ void unit_issue(floatx16& X, const floatx16& cb, const half8 (&ah)[KS], const half8 (&bq)[KS]) {
    asm volatile("v_mfma_f32_32x32x16_f16 %[X], %[h0], %[q0], %[cb]\n\t" : [X] "=&v"(X) : [h0] "v"(ah[0]), [q0] "v"(bq[0]), [cb] "v"(cb));
#pragma unroll
    for (int s = 1; s < KS; ++s)
        asm volatile("v_mfma_f32_32x32x16_f16 %[X], %[h], %[q], %[X]\n\t" : [X] "+v"(X) : [h] "v"(ah[s]), [q] "v"(bq[s]));
}

}  // namespace sknnr
