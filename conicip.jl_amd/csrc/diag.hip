// Diagonal-block kernel of the blocked LDL' (v2): in-LDS LDL' of one 128x128 block, one workgroup of NW waves (8 by
// default), micro-blocked by 16 with v_mfma_f64_16x16x4_f64 -- and the launches built around it: the standalone kernel
// (k_ldlt_diag128_v2), diagonal kernel + previous panel's in-block update (k_ldlt_diag_upd), and the one launch per panel
// of the serial schedule (k_ldlt_panel: + this panel's TRSM, pipelined behind the diagonal kernel).
//
// This is the serial link of the factorisation chain (N/128 of these run back to back), so it is
// organised around latency, not throughput:
//   for each 16-column micro-panel kb:
//     A. wave 0: LDL' of the 16x16 diagonal micro-block + inverse of its unit-lower factor, the tile held in the MFMA
//        accumulator layout, pivots broadcast with v_readlane, pivot row / column entries moved with ds_bpermute
//     B. every working wave, for its row tiles below: W = U * inv(L11)' as 4 MFMAs (the accumulator layout of
//        f64 16x16x4 is also its operand layout); L = W D^-1 written back to the LDS image
//     C. the helper waves: trailing tiles C[it][jt] -= (L[it] D) L[jt]' (4 MFMAs per 16x16 tile, three tiles at a time per
//        wave) and the write-back of micro-panel kb -- while wave 0 is already in step A of micro-panel kb+1
//   The explicit inverses of the 128x128 unit-lower factors (X = inv(L) by block rows, a chain of MFMAs whose running tiles
//   stay in registers) are not on the chain: one batched launch after the factorisation (k_diag_inverse_batched).
//
// LDS image: a[row + col*144] (pitch 144 doubles: fragment reads with rows on lanes 0-15 and k on the
// lane groups are bank-conflict free), the 8 micro inverses, and d / 1/d in the pitch padding:
// 160 KiB exactly (one workgroup per CU, which is all a serial kernel needs).
#include "cip_internal.h"
#include "cip_gemm_tile.h"
#include <stdlib.h>

#define DP 144
#ifndef DIAG_SKIP
#define DIAG_SKIP 0        // development switch (tools/diag_bench.hip): bit mask of phases to skip
#endif
#ifdef DIAG_TIMING
__device__ long g_diag_t[8 * 8];                 // [kb][slot]: s_memtime stamps (tools/diag_bench.hip)
#define DIAG_STAMP(kb, slot, cond) do { if (cond) g_diag_t[(kb) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#ifndef DIAG_TIMING_TID
#define DIAG_TIMING_TID 64                       // first lane of the helper wave whose phases are stamped (wave 1; 128, 192, 320, 384, 448 for the others)
#endif
#else
#define DIAG_STAMP(kb, slot, cond) do { } while (0)
#endif
#ifdef PANEL_TIMING
// development (tools/panel_stamps.py): s_memrealtime (100 MHz, one clock for the chip) at the phases of a k_ldlt_panel<true> launch
__device__ long g_panel_t[32];
__device__ int g_panel_col = -1;                 // -1: every launch stamps (the last one stays); else the launch of the panel at this column
#define PANEL_STAMP(slot, cond) do { if ((cond) && (g_panel_col < 0 || g_panel_col == col0)) g_panel_t[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int cip_debug_panel_stamps(long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_panel_t), sizeof(long) * 32); }
extern "C" int cip_debug_panel_col(int col) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_panel_col), &col, sizeof(int)); }
#else
#define PANEL_STAMP(slot, cond) do { } while (0)
#endif
#define XM_OFF (CIP_NB * DP)             // 8 x 256 doubles: xm[kb][k*16 + jj] = Xm_kb[jj][k]
#define DIAG2_LDS_BYTES ((CIP_NB * DP + 8 * 256) * 8)

__device__ __forceinline__ double rlane(double x, int lane) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    return r;
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ double shfl_d(double x, int src) { return __shfl(x, src, 64); }
// x of lane (byte_addr / 4): ds_bpermute with a precomputed byte address (the per-pivot part of the source lane is a
// multiple of 16 lanes = 64 bytes and folds into the instruction's offset field)
__device__ __forceinline__ double bperm_d(double x, int byte_addr) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_ds_bpermute(byte_addr, lo);
    hi = __builtin_amdgcn_ds_bpermute(byte_addr, hi);
    return __hiloint2double(hi, lo);
}

// Step A: LDL' of the 16x16 diagonal micro-block kb and the inverse X of its unit-lower factor, by one wave.
//
// Round 3 form.  The 16 pivots are taken in four BLOCKS OF FOUR columns, and the tile is held so that a block never needs
// a cross-lane VECTOR move while it is being factored:
//   * lane (l15, g), register q holds tile element (row P(l15), column 4g + q), P(x) = 4 (x mod 4) + x div 4 -- the f64-MFMA
//     accumulator layout with rows AND columns relabelled by P (the hardware pairs register q of lane group g with operand
//     lane g + 4q = P(4g + q); relabelling both sides by the same involution keeps every MFMA consistent).  Block jb = the
//     four registers of lane group jb: row-per-lane, 16 lanes.  X likewise: lane (l15, g) register q = X[4g + q][P(l15)].
//   * inside block jb (EXEC = lane group jb) a pivot j is: d = v_readlane, 1/d (rcp + two Newton steps), the multipliers
//     l = column * (1/d) (one multiply), and for each later column k of the block the SCALAR a[k][j] by v_readlane and one
//     fma (SGPR operand) on the column, one multiply + one fma on X's row: no ds_bpermute, no select.
//   * the block's effect on everything to its right is ONE rank-4 update each for the tile and for X,
//         u[:, k] += sum_kk (-l_kk)[:] * a[k][j_kk]        X[r, :] += sum_kk (-l[r][j_kk]) * X[j_kk][:],
//     two v_mfma_f64_16x16x4_f64 in place on the register tuples.  Their operands want pivot kk's vector in lane group
//     kk: the block's lanes park their four l / unnormalised-column / X-row registers in LDS (the not-yet-written slot of
//     this micro-panel's inverse) and every lane reads back the one of its group -- 12 narrow ds_write_b64 + 3 ds_read_b64
//     per block instead of 18 ds_bpermute_b32 strung along the pivots.  Operand lanes of finished columns / rows read
//     zeros, so the in-place MFMA leaves them alone (x + 0 * y).
// v_mfma_f64_16x16x4_f64 IS the chain acc <- fma(a_kk, b_kk, acc), kk = 0, 1, 2, 3, each step rounded (tools/mfma_order.hip:
// 512 000 random cases, no mismatch against that order, 30-40 % against any other), so every stored entry receives the
// same fused multiply-adds on the same operands in the same order as in round 2's pivot-by-pivot form: L, d, 1/d and the
// micro inverse are BIT-IDENTICAL to it (tools/diag_bench.hip / tools/stepa_probe.hip build that form with
// -DDIAG_STEP_A_REF for the comparison).
// What was measured on the way (tools/stepa_probe.hip, one wave alone on a CU, clocks per 16 pivots): round 2's form 4280
// (267 per pivot); the same arithmetic blocked by four but still in the plain accumulator layout (three ds_bpermute pairs
// per pivot, selects) 4110 -- of which the permutes 1400 (every ds_bpermute_b32 occupies the LDS pipe for ~17 clocks and
// the last pair of a pivot is on its chain), the three MFMA rounds 1200 (register-tuple copies around them), the Newton
// steps 500; the dependency chain of a pivot alone (readlane, rcp, two Newton steps, multiply, fma) is 80.
// (Round 2 variants: row-per-lane for the whole tile 40.5 us per kernel against 30.9; every operand through
// v_permlane*_swap / DPP instead of ds_bpermute 31.2 against 30.6.)
__host__ __device__ constexpr int diag_perm(int x) { return 4 * (x & 3) + (x >> 2); }
// x of lane N of the own 16-lane row (DPP row_newbcast: gfx90a and later)
template <int N>
__device__ __forceinline__ double row_bcast_t(double x) {
    return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + N, 0xf, 0xf, true);      // v_mov_b64_dpp (gfx90a+: 64-bit DPP knows row_newbcast only)
}
// acc += x[lane N of the own row] * (-y) as ONE instruction: v_fmac_f64 is a VOP2 and takes DPP on its first source (the
// compiler keeps v_mov_b64_dpp + v_fma_f64: two instructions on a wave that pays ~10 clocks per instruction).  The DPP
// source must not have been written by the instruction right before (VALU write -> DPP read: 2 wait states; the
// assembler does not check inline asm): in diag_step_a it is always a register written at least four instructions earlier.
template <int N>
__device__ __forceinline__ void fmac_bcast_neg_t(double &acc, double x, double y) {
    asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(y), "n"(N));
}
// One block of four pivots, every lane index a TEMPLATE constant: with run-time-looking loop indices the optimiser merged the
// four structurally identical EXEC-masked regions of diag_step_a, turned the DPP lane selects into PHIs and the selection
// into compare-and-branch ladders (35.7 us per kernel instead of 21.8).
template <int JB, int GJ, int GK>
__device__ __forceinline__ void stepa_update(v4d &T, const v4d &S, const v4d &M) {     // T[gk] -= S[gj] @ (lane of row 4JB + gk) * M[gj], gk = GK .. 3
    if constexpr (GK < 4) {
        double t = T[GK];
        fmac_bcast_neg_t<diag_perm(4 * JB + GK)>(t, S[GJ], M[GJ]);
        T[GK] = t;
        stepa_update<JB, GJ, GK + 1>(T, S, M);
    }
}
template <int JB, int GJ>
__device__ __forceinline__ void stepa_pivots(v4d &U, v4d &L, double (&dd)[4]) {
    if constexpr (GJ < 4) {
        // the block lives in ONE 16-lane row: a[j][j] and a[k][j] reach the row's lanes by DPP row broadcast (v_mov_b64_dpp /
        // v_fmac_f64_dpp, VALU latency) instead of v_readlane -> SGPR -> VALU (28 clocks on every pivot's chain)
        const double d = row_bcast_t<diag_perm(4 * JB + GJ)>(U[GJ]);
#ifdef DIAG_NEWTON2
        const double rho = fast_rcp(d);
#else
        // Round 5: ONE Newton step on the chain.  A lone wave pays 6 clocks per fp64 instruction whether it depends on its
        // predecessor or not, and 17 for v_rcp_f64 (tools/issue_probe.hip): the reciprocal with two Newton steps was 47 of a
        // pivot's ~150 clocks.  v_rcp_f64 is good to 2^-25, one step leaves <= 11 ulp (two are correctly rounded;
        // tools/rcp_probe.hip) -- and the factor stays backward stable because the STORED pivot is defined from the reciprocal
        // that was used, not the other way round: every multiplier is l = u * rho and every update subtracts u_k u_i rho, so what
        // was computed is exactly consistent with d := 1 / rho, which is what is stored (two Newton steps, off the chain, at the end
        // of the micro-block), together with rho itself as 1/d.  Against the textbook pivot that d is a relative perturbation of
        // the matrix's own diagonal entry a[j][j] of <= 11 ulp.  Different rounding than rounds 2-4 (the chain forms of the library
        // share this code and stay bit-identical to each other).  -DDIAG_NEWTON2 builds the old form for the A/B tools.
        double rho = __builtin_amdgcn_rcp(d);
        rho = fma(rho, fma(-d, rho, 1.0), rho);
#endif
        dd[GJ] = rho;                                  // kept for the output: the in-place MFMAs turn a NaN multiplier into NaNs in
                                                       // FINISHED columns too (0 * NaN), and the first bad pivot must be reported at its own column
        L[GJ] = U[GJ] * rho;                           // multipliers l_ij = a_ij * rho_j (the diagonal lane: never stored)
        stepa_update<JB, GJ, GJ + 1>(U, U, L);         // U[gk] -= a[k][j] * l_ij, a[k][j] (k = 4JB + gk) from the row's lane that holds row k
        stepa_pivots<JB, GJ + 1>(U, L, dd);
    }
}
template <int JB, int GJ>
__device__ __forceinline__ void stepa_xrows(v4d &X, const v4d &L) {
    if constexpr (GJ < 3) {
        stepa_update<JB, GJ, GJ + 1>(X, L, X);         // X[k][:] -= l[k][j] X[j][:], l[k][j] = a[k][j] * (1/d_j) = L[gj] of the row's lane that holds row k
        stepa_xrows<JB, GJ + 1>(X, L);
    }
}
#ifdef STEPA_TIMING
__device__ long g_stepa_t[32];
#define STEPA_STAMP(i) do { if (threadIdx.x == 0) g_stepa_t[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STEPA_STAMP(i) do { } while (0)
#endif

#ifdef DIAG_STEPA_PAIR
// ---------------------------------------------------------------------------------------------------------------------
// A/B form (round 5, NOT the default: it lost): TWO COLUMNS PER CHAIN LINK.  Inside a block of four the pivots are taken in
// pairs (j, j + 1).  With a = a[j][j], b = a[j+1][j], c = a[j+1][j+1] (three DPP row broadcasts, independent of each other) the
// second pivot of the pair is the Schur complement c - b^2 / a = det / a, det = a c - b^2, so its reciprocal is a * (1 / det) --
// it does not wait for 1/a: the two reciprocal chains run side by side.  Built on the assumption that the serial wave is bound
// by its dependency chain; tools/issue_probe.hip says a lone wave pays 6 clocks per fp64 instruction dependent or not, and this
// form has MORE instructions: 808 against 616 ticks per block of four pivots, step A 4170 against 3960 per micro-block
// (tools/stepa_probe.hip built with -DDIAG_STEPA_PAIR; residual 1.6e-16).  Kept for the probe only.
template <int N>
__device__ __forceinline__ void fmac_bcast_neg_s(double &acc, double x, double y) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(y), "n"(N));
}
#define PAIR_FMAC(N, ACC, X_, Y_) do { double t_ = ACC; fmac_bcast_neg_s<N>(t_, X_, Y_); ACC = t_; } while (0)
template <int JB, int GJ>
__device__ __forceinline__ void stepa_pair(v4d &U, v4d &X, v4d &L, double (&rho)[4]) {
    constexpr int LA = diag_perm(4 * JB + GJ), LB = diag_perm(4 * JB + GJ + 1);      // lanes (of the 16-lane row) that hold rows j, j + 1
    const double a = row_bcast_t<LA>(U[GJ]);
    const double b = row_bcast_t<LB>(U[GJ]);
    const double c = row_bcast_t<LB>(U[GJ + 1]);
    const double det = fma(a, c, -(b * b));
    double rdet = __builtin_amdgcn_rcp(det);
    rdet = fma(rdet, fma(-det, rdet, 1.0), rdet);
    double ra = __builtin_amdgcn_rcp(a);
    ra = fma(ra, fma(-a, ra, 1.0), ra);
    const double r2 = a * rdet;
    rho[GJ] = ra;
    rho[GJ + 1] = r2;
    L[GJ] = U[GJ] * ra;
    U[GJ + 1] = fma(-b, L[GJ], U[GJ + 1]);                     // column j + 1 as the later columns see it (b is already in every lane)
    L[GJ + 1] = U[GJ + 1] * r2;
    X[GJ + 1] = fma(-(b * ra), X[GJ], X[GJ + 1]);              // X[j+1][:] -= l[j+1][j] X[j][:]
    if constexpr (GJ == 0) {
        PAIR_FMAC(diag_perm(4 * JB + 2), U[2], U[0], L[0]);
        PAIR_FMAC(diag_perm(4 * JB + 3), U[3], U[0], L[0]);
        PAIR_FMAC(diag_perm(4 * JB + 2), U[2], U[1], L[1]);
        PAIR_FMAC(diag_perm(4 * JB + 3), U[3], U[1], L[1]);
        PAIR_FMAC(diag_perm(4 * JB + 2), X[2], L[0], X[0]);
        PAIR_FMAC(diag_perm(4 * JB + 3), X[3], L[0], X[0]);
        PAIR_FMAC(diag_perm(4 * JB + 2), X[2], L[1], X[1]);
        PAIR_FMAC(diag_perm(4 * JB + 3), X[3], L[1], X[1]);
    }
}
#endif
template <int JB>
__device__ __forceinline__ void stepa_round(v4d &U, v4d &X, v4d &L, double (&dd)[4], double *scr, int l15, int g) {
    STEPA_STAMP(1 + 4 * JB);
    if (g == JB) {
#ifdef DIAG_STEPA_PAIR
        stepa_pair<JB, 0>(U, X, L, dd);
        stepa_pair<JB, 2>(U, X, L, dd);
        if (JB < 3) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { scr[kk * 16 + l15] = L[kk]; scr[64 + kk * 16 + l15] = U[kk]; scr[128 + kk * 16 + l15] = X[kk]; }
        }
#else
        stepa_pivots<JB, 0>(U, L, dd);
        if (JB < 3) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                scr[kk * 16 + l15] = L[kk];
                scr[64 + kk * 16 + l15] = U[kk];               // column 4JB + kk as the later columns see it (before the division)
            }
        }
        // the same multipliers on the block's own rows of X (under the LDS latency of the stores above)
        stepa_xrows<JB, 0>(X, L);
        if (JB < 3) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) scr[128 + kk * 16 + l15] = X[kk];
        }
#endif
    }
    STEPA_STAMP(2 + 4 * JB);
    if (JB < 3) {
        // acc[q] @ lane (l15, g)  +=  sum_kk R @ lane (l15, kk) * P @ lane (g + 4q, kk).  Lanes exchange data through LDS
        // here -- ONE round trip per block for the three operand sets: the hardware completes a wave's DS operations in
        // order, so a read issued behind the writes sees them -- but the COMPILER must be told that other lanes' stores
        // matter (without the wavefront-scope release / acquire pair it sank two of the three reads into the block's
        // EXEC-masked region: 12 lane groups read stale data)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool todo = (l15 & 3) > JB;                      // hardware index l15 <-> column / X row 4 (l15 & 3) + (l15 >> 2): beyond the block?
        const double ln = -scr[g * 16 + l15];
        const double pu = scr[(todo ? 64 + g * 16 : 192) + l15];
        const double xr = scr[128 + g * 16 + l15];
        STEPA_STAMP(3 + 4 * JB);
        U = MFMA(pu, ln, U);                                   // the tile first: the next block's first pivot waits for it
        X = MFMA(todo ? ln : 0.0, xr, X);
        STEPA_STAMP(4 + 4 * JB);
    }
}
#ifndef DIAG_STEP_A_REF
// Round 5: the serial wave's step C -- the last rank-16 update of the NEXT diagonal micro-block, C -= (L D) L' with L = the
// tile's own rows of micro-panel kb -- computed straight into step A's register layout (lane (l15, g), register q = element
// (row P(l15), column 4g + q)): the operand lanes read row P(l15) instead of row l15, the MFMAs are the same four on the same
// numbers in the same k order (bit-identical to diag_step_c), and the tile never makes the LDS round trip between C and A.
__device__ __forceinline__ v4d diag_step_c_perm(const double *a, int kb, int l15, int g, const double (&d4)[4]) {
    const int c = kb * 16, t0 = (kb + 1) * 16, row = diag_perm(l15);
    const double *cp = a + (t0 + row) + (t0 + 4 * g) * DP;
    const double *pl = a + (t0 + row) + (c + g) * DP;
    v4d acc = (v4d){cp[0], cp[DP], cp[2 * DP], cp[3 * DP]};
    double v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = pl[4 * s * DP];
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = MFMA(v[s], -(v[s] * d4[s]), acc);
    return acc;
}
// Round 5: the serial wave's B and C in one piece.  Its B tile is L21 = the NEXT diagonal micro-block's rows of micro-panel kb
// (W = U21 inv(L11)', L21 = W D^-1: 4 MFMAs), and its C step needs exactly that tile as BOTH operands (C11 -= (L21 D) L21'):
// the accumulator layout of v_mfma_f64_16x16x4_f64 is also its operand layout, so L21 goes from B's accumulator registers
// straight into C's MFMAs -- written to LDS (for the helpers, who read it after the B count) but never read back -- and the
// C11 tile's loads are issued in front of B.  Rows permuted by P throughout (see diag_step_c_perm): the result is step A's
// register tile.  Same MFMAs on the same numbers in the same order as diag_step_b + diag_step_c: same bits.
struct BcOperands { double u[4]; v4d c; };
__device__ __forceinline__ BcOperands diag_step_bc_load(const double *a, int kb, int l15, int g) {
    const int c = kb * 16, t0 = (kb + 1) * 16, row = diag_perm(l15);
    const double *pl = a + (t0 + row) + (c + g) * DP;
    const double *cp = a + (t0 + row) + (t0 + 4 * g) * DP;
    BcOperands o;
#pragma unroll
    for (int s = 0; s < 4; ++s) o.u[s] = pl[4 * s * DP];
    o.c = (v4d){cp[0], cp[DP], cp[2 * DP], cp[3 * DP]};
    return o;
}
__device__ __forceinline__ v4d diag_step_bc_perm(double *a, int kb, int l15, int g, const BcOperands &o, const double (&xa)[4],
                                                 const double (&di4)[4], const double (&d4)[4]) {
    const int c = kb * 16, t0 = (kb + 1) * 16, row = diag_perm(l15);
    double *pl = a + (t0 + row) + (c + g) * DP;
    v4d acc2 = o.c;
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = MFMA(xa[s], o.u[s], acc);
    double v[4], w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] = acc[q] * di4[q]; pl[4 * q * DP] = v[q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = -(v[q] * d4[q]);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc2 = MFMA(v[s], w[s], acc2);
    return acc2;
}
template <bool PRELOADED = false>
__device__ __forceinline__ void diag_step_a(double *a, double *xm, int kb, int lane, int *info, int col0, PivotSigns sg,
                                            v4d U0 = (v4d){0.0, 0.0, 0.0, 0.0}) {
    const int l15 = lane & 15, g = lane >> 4;
    const int c = kb * 16;
    const int row = diag_perm(l15);                            // the tile row / X column this lane holds
    v4d U, X, L = (v4d){0.0, 0.0, 0.0, 0.0};
    double dd[4] = {0.0, 0.0, 0.0, 0.0};                       // lane group jb: the reciprocals rho_j of the four pivots of its block
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        U[q] = PRELOADED ? U0[q] : a[(c + row) + (c + 4 * g + q) * DP];
        X[q] = (4 * g + q == row) ? 1.0 : 0.0;
    }
    double *scr = xm + kb * 256;                               // [set][kk][l15], three sets: this slot is written at the very end
    // operand lanes of the rank-4 updates: lane (l15, kk) carries the vector of pivot kk; P-side lanes whose hardware
    // index belongs to a finished / current block ((l15 & 3) <= jb) must contribute zeros -> they read the zeroed 4th set
    if (lane < 16) scr[192 + lane] = 0.0;
    STEPA_STAMP(0);
    if (!(DIAG_SKIP & 8)) {
        stepa_round<0>(U, X, L, dd, scr, l15, g);
        stepa_round<1>(U, X, L, dd, scr, l15, g);
        stepa_round<2>(U, X, L, dd, scr, l15, g);
        stepa_round<3>(U, X, L, dd, scr, l15, g);
    }
    STEPA_STAMP(17);
    double rsel = 0.0;                                         // the reciprocal the lane's diagonal pivot was used with
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = 4 * g + q;
        if (col == row) rsel = dd[q];
    }
    const double dsel = fast_rcp(rsel);                        // the stored pivot d := 1 / rho (correctly rounded in all but 1 of 4000 cases)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = 4 * g + q;
        const double v = (col == row) ? dsel : L[q];           // d on the diagonal, l_ij below (above: not stored)
        if (col <= row) a[(c + row) + (c + col) * DP] = v;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) xm[kb * 256 + row * 16 + 4 * g + q] = X[q];      // xm[k = cc][jj = r] = X[r][cc]
    {
        // lane (l15, g = l15 & 3) holds the diagonal element of column `row` (= 4g + (l15 >> 2)); these 16 lanes are in
        // column order, so the lowest set bit of a ballot is the FIRST bad column of the micro-block (one atomic per flag
        // instead of one per bad lane racing for the word)
        const bool diag_lane = g == (l15 & 3);
        // a bad pivot: zero, non-finite, or -- the matrix is quasi-definite in this static order -- of the wrong sign
        const int col = col0 + c + row;
        const bool want_pos = (col >= sg.p0 && col < sg.p1) || col >= sg.N;
        // info[0]: first bad pivot of any kind; info[2]: first zero / non-finite one (fatal even for a regularised factor,
        // whose wrong-sign pivots the refinement of solve3x3 absorbs)
        const bool dead = diag_lane && !(fabs(dsel) > 0.0 && fabs(dsel) < 1.7e308);
        const bool bad = dead || (diag_lane && sg.p0 >= 0 && (dsel > 0.0) != want_pos);
        const unsigned long long mbad = __ballot(bad), mdead = __ballot(dead);
        if (mbad && lane == __ffsll((long long)mbad) - 1) atomicCAS(info, 0, col + 1);
        if (mdead && lane == __ffsll((long long)mdead) - 1) atomicCAS(info + 2, 0, col + 1);
        if (diag_lane) {
            a[128 + (c + row) * DP] = dsel;
            a[129 + (c + row) * DP] = rsel;
        }
    }
}
#else
// Round 2's form (pivot by pivot over the whole tile, plain accumulator layout: lane (row l15, group g) owns columns g,
// g+4, g+8, g+12; six ds_bpermute pairs and up to eight selected updates per pivot) -- kept for the A/B tools only.
__device__ __forceinline__ void diag_step_a(double *a, double *xm, int kb, int lane, int *info, int col0, PivotSigns sg) {
    const int l15 = lane & 15, g = lane >> 4;
    const int c = kb * 16;
    double u[4], x[4];
    int addr_c[4];                                             // byte address of lane ((g + 4q) & 15) of lane group 0
    const int addr_r = l15 * 4;                                // ... of lane l15 of lane group 0
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u[q] = a[(c + l15) + (c + g + 4 * q) * DP];
        x[q] = (g + 4 * q == l15) ? 1.0 : 0.0;
        addr_c[q] = ((g + 4 * q) & 15) * 4;
    }
#pragma unroll
    for (int j = 0; j < ((DIAG_SKIP & 8) ? 0 : 16); ++j) {
        const int gj = j & 3, qj = j >> 2;
        const double d = rlane(u[qj], 16 * gj + j);
        const double wi = bperm_d(u[qj], addr_r + 64 * gj);    // a[i][j], own row
        const double xj = bperm_d(x[qj], addr_r + 64 * gj);    // X[j][cc], own column
        double cjv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)                            // a[g+4q][j] where some lane group still has column g+4q > j
            cjv[q] = (4 * q + 3 > j) ? bperm_d(u[qj], addr_c[q] + 64 * gj) : 0.0;
        const double di = fast_rcp(d);
        const double ti = wi * di;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (4 * q + 3 > j) {
                const double nu = u[q] - ti * cjv[q];
                const double nx = x[q] - (cjv[q] * di) * xj;
                if (4 * q > j) { u[q] = nu; x[q] = nx; }
                else {
                    const bool act = g + 4 * q > j;
                    u[q] = act ? nu : u[q];
                    x[q] = act ? nx : x[q];
                }
            }
        }
        if (g == gj) u[qj] = (l15 == j) ? d : ti;              // column j final: l_ij below, d on the diagonal
    }
    double dsel = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = g + 4 * q;
        if (col <= l15) a[(c + l15) + (c + col) * DP] = u[q];
        if (col == l15) dsel = u[q];
        xm[kb * 256 + l15 * 16 + col] = x[q];                  // xm[k = cc][jj = r] = X[r][cc]
    }
    if (g == (l15 & 3)) {
        const int col = col0 + c + l15;
        const bool want_pos = (col >= sg.p0 && col < sg.p1) || col >= sg.N;
        const bool dead = !(fabs(dsel) > 0.0 && fabs(dsel) < 1.7e308);
        if (dead || (sg.p0 >= 0 && (dsel > 0.0) != want_pos)) atomicCAS(info, 0, col + 1);
        if (dead) atomicCAS(info + 2, 0, col + 1);
        a[128 + (c + l15) * DP] = dsel;
        a[129 + (c + l15) * DP] = fast_rcp(dsel);
    }
}
#endif

// step B for one row tile: W = U inv(L11)' (4 MFMAs), L = W D^-1 written back into the LDS image
__device__ __forceinline__ void diag_step_b(double *a, int it, int c, int l15, int g, const double (&xa)[4],
                                            const double (&di4)[4]) {
    double *p = a + (it * 16 + l15) + (c + g) * DP;
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = MFMA(xa[s], p[4 * s * DP], acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) p[4 * q * DP] = acc[q] * di4[q];
}
// step C for one tile (it, jt): C -= (L[it] D) L[jt]', both operands from the LDS image (any wave can take any tile)
__device__ __forceinline__ void diag_step_c(double *a, int it, int jt, int c, int l15, int g, const double (&d4)[4]) {
    double *cp = a + (it * 16 + l15) + (jt * 16 + g) * DP;
    const double *lj = a + (jt * 16 + l15) + (c + g) * DP;
    const double *li = a + (it * 16 + l15) + (c + g) * DP;
    v4d acc = (v4d){cp[0], cp[4 * DP], cp[8 * DP], cp[12 * DP]};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = MFMA(lj[4 * s * DP], -(li[4 * s * DP] * d4[s]), acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) cp[4 * q * DP] = acc[q];
}

// step C for up to three tiles at once (tiles idx, idx + st, idx + 2 st of step kb in row-major order over rows
// it = kb+2..7, jt = kb+1..it): every operand read is issued before the first MFMA and the three accumulation chains
// interleave -- one tile at a time is an LDS round trip plus four DEPENDENT 64-cycle MFMAs, and the compiler cannot
// overlap consecutive tiles (it must assume that a tile's write-back aliases the next tile's reads).  The helper waves'
// tiles were the diagonal kernel's critical path at the first micro-panels (tools/diag_bench: 30.4 -> 23.8 us with the
// tiles switched off).  Same MFMAs on the same operands: identical results.
__device__ __forceinline__ void diag_tile_of(int kb, int idx, int &it, int &jt) {
    int r = 0;
    while ((r + 1) * (r + 4) / 2 <= idx) ++r;                // row r holds r + 2 tiles, r (r + 3) / 2 before it
    it = kb + 2 + r;
    jt = kb + 1 + idx - r * (r + 3) / 2;
}
template <int NT>
__device__ __forceinline__ void diag_step_c_multi(double *a, int kb, int idx, int st, int c, int l15, int g, const double (&d4)[4]) {
    double *cp[NT];
    v4d acc[NT];
    double lj[NT][4], li[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int it, jt;
        diag_tile_of(kb, idx + st * t, it, jt);
        cp[t] = a + (it * 16 + l15) + (jt * 16 + g) * DP;
        const double *pj = a + (jt * 16 + l15) + (c + g) * DP, *pi = a + (it * 16 + l15) + (c + g) * DP;
        acc[t] = (v4d){cp[t][0], cp[t][4 * DP], cp[t][8 * DP], cp[t][12 * DP]};
#pragma unroll
        for (int s = 0; s < 4; ++s) { lj[t][s] = pj[4 * s * DP]; li[t][s] = pi[4 * s * DP]; }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = MFMA(lj[t][s], -(li[t][s] * d4[s]), acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) cp[t][4 * q * DP] = acc[t][q];
}

// Round 3: the helper waves' trailing tiles LEFT-LOOKING.  Tile (it, jt) receives ALL its steps 0 .. n-1 in one visit,
//     C -= sum_s (L[it][s] D_s) L[jt][s]',    the accumulator in registers across the steps,
// just before it is needed (column kb + 1 and the next diagonal tile during step kb), instead of one visit per step for
// every tile of the trailing matrix.  Per tile the same MFMAs on the same operands in the same order as the step-by-step
// form (a store and a reload of the accumulator between steps change nothing): the factor is bit-identical.  What changes
// is WHEN the work is done: step by step the helpers had 27 / 20 / 14 / 9 / 5 / 2 tiles to visit in steps 0 .. 5 and the
// serial wave waited for them at the first two or three (tools/diag_bench -DDIAG_TIMING: 8230 / 6750 / 5200 clocks per
// micro-panel against 4900 once the helpers keep up); left-looking it is 7 / 6 / 5 / 4 / 3 / 2 tiles with 1 .. 6 steps each
// (at most one C round trip per tile, the next step's operands fetched under the current step's MFMAs), well inside the
// serial wave's 3900 clocks at every step.
__device__ __forceinline__ void diag_tile_left(double *a, int it, int jt, int nsteps, int l15, int g) {
    double *cp = a + (it * 16 + l15) + (jt * 16 + g) * DP;
    const double *pj = a + (jt * 16 + l15) + g * DP, *pi = a + (it * 16 + l15) + g * DP, *pd = a + 128 + g * DP;
    // Round 5: TWO accumulation chains.  With one accumulator and fresh operands per MFMA a step's four MFMAs cost ~100 clocks each
    // (tools/diag_bench -DDIAG_TIMING; back to back on fixed registers they issue every 64, tools/issue_probe.hip), and from
    // micro-panel 3 on the helpers, not the serial wave, bound the diagonal kernel (3100 ticks of tiles at kb = 3 beside 3850 of
    // C + A, and the helpers also store the panel).  Now the k-slices alternate between acc (which starts from C) and acc2
    // (which starts from zero), C' = acc + acc2 at the end: 3100 -> 2100 ticks.  A different summation order than rounds 3-4 --
    // the same for every tile of every chain form of the library, all of which run this code.
    v4d acc = (v4d){cp[0], cp[4 * DP], cp[8 * DP], cp[12 * DP]};
    v4d acc2 = (v4d){0.0, 0.0, 0.0, 0.0};
    double lj[4], li[4], dv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { lj[k] = pj[4 * k * DP]; li[k] = pi[4 * k * DP]; dv[k] = pd[4 * k * DP]; }
    for (int s = 0; s < nsteps; ++s) {
        double nj[4], ni[4], nd[4];
        const int o = (s + 1 < nsteps ? 16 * (s + 1) : 16 * s) * DP;      // (the last step re-reads its own operands: no branch around the loads)
#pragma unroll
        for (int k = 0; k < 4; ++k) { nj[k] = pj[o + 4 * k * DP]; ni[k] = pi[o + 4 * k * DP]; nd[k] = pd[o + 4 * k * DP]; }
        acc = MFMA(lj[0], -(li[0] * dv[0]), acc);
        acc2 = MFMA(lj[1], -(li[1] * dv[1]), acc2);
        acc = MFMA(lj[2], -(li[2] * dv[2]), acc);
        acc2 = MFMA(lj[3], -(li[3] * dv[3]), acc2);
#pragma unroll
        for (int k = 0; k < 4; ++k) { lj[k] = nj[k]; li[k] = ni[k]; dv[k] = nd[k]; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) cp[4 * q * DP] = acc[q] + acc2[q];
}

// X block row `it` (runtime, wave-uniform): tiles kept in registers, statically indexed
// P: pitch of the block image; XR: pitch of a row of a micro inverse in `xm` (tile = 16 XR doubles).  The accesses here run along
// COLUMNS of the image (lane index l15 -> column): with the diagonal kernel's pitch 144 (288 dwords = 32 mod 64) sixteen lanes of a
// half-wave shared four banks -- PMC: 0.88 LDS bank-conflict cycles per LDS-active cycle in k_diag_inverse_batched; with P = 130
// (260 dwords = 4 mod 64) and XR = 17 the same reads are conflict-free (tools/pmc_sq.sh).
template <int P, int XR>
__device__ __forceinline__ void diag_inverse_row(double *a, const double *xm, int it, int l15, int g) {
    double XT[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) XT[t][s] = 0.0;
    // X[it][it] = Xm[it]: element (row l15, col g+4s)
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t == it) {
#pragma unroll
            for (int s = 0; s < 4; ++s) XT[t][s] = xm[t * (16 * XR) + (g + 4 * s) * XR + l15];
        }
#pragma unroll
    for (int jt = 6; jt >= 0; --jt) {
        if (jt < it) {
            v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kt = 7; kt >= 1; --kt) {
                if (kt > jt && kt <= it) {
                    // Aop[cjt][k] = L[kt*16 + k][jt*16 + cjt]
                    const double *lp = a + (kt * 16 + g) + (jt * 16 + l15) * P;
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = MFMA(lp[4 * s], XT[kt][s], acc);
                }
            }
            // X[it][jt] = -S * Xm[jt]:  Aop[c'][k] = Xm[jt][k][c'] = xm[jt][c'*16 + k]
            const double *xp = xm + jt * (16 * XR) + l15 * XR + g;
            v4d r = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) r = MFMA(xp[4 * s], acc[s], r);
#pragma unroll
            for (int s = 0; s < 4; ++s) XT[jt][s] = -r[s];
        }
    }
    // park the finished block row in the (now free) upper triangle: X[r][cc] (r > cc) -> a[cc + r*DP]
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t < it) {
#pragma unroll
            for (int s = 0; s < 4; ++s) a[(t * 16 + g + 4 * s) + (it * 16 + l15) * P] = XT[t][s];
        }
}

// 128x128 block <-> LDS image, 16-byte accesses, all loads of a thread in flight before the first store
template <int P = DP>
__device__ __forceinline__ void diag_load_block(double *a, const double *Kb, long ld, int tid) {
    v2d tmp[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const int e = q * 256 + tid;             // pair index: rows 2*(e&63), +1 ; column e>>6
        tmp[q] = *(const v2d *)(Kb + 2 * (e & 63) + (long)(e >> 6) * ld);
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const int e = q * 256 + tid;
        const int i = 2 * (e & 63), j = e >> 6;
        v2d v = tmp[q];
        if (i < j) v.x = 0.0;                    // strictly upper part -> 0
        if (i + 1 < j) v.y = 0.0;
        *(v2d *)(a + i + j * P) = v;
    }
}

// columns [c, c+16) of the LDS image -> K (L strictly lower, d on the diagonal), 16-byte stores where aligned pairs
// lie below the diagonal; `t` of `nt` threads.  PUB: write-through (`sc1`) stores -- the readers are other workgroups of
// the same launch (k_ldlt_panel's TRSM strips)
#ifndef DIAG_OLD_STAGE_COUNT
#define PANEL_STAGE_LAST 0x10000u            // the serial wave's count on a panel launch's stage word (see stage_reached)
#else
#define PANEL_STAGE_LAST ((unsigned)NH)      // rounds 5 - 6a (tools/stage_mix_demo.sh): one running count for both kinds
#endif
__device__ __forceinline__ void st_pub(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_pub(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool PUB = false>
__device__ __forceinline__ void diag_store_panel(const double *a, double *Kb, long ld, int c, int t, int nt) {
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)Kb, 0, 0x7fffffff, 0x00020000);
    for (int e = t; e < 16 * 64; e += nt) {
        const int i = 2 * (e & 63), j = c + (e >> 6);
        if (i >= j) {
            const v2d v = *(const v2d *)(a + i + j * DP);
            if (PUB) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_t, v), rs, (int)((i + (long)j * ld) * 8), 0, 16);
            else *(v2d *)(Kb + i + (long)j * ld) = v;
        } else if (i + 1 == j) {
            if (PUB) st_pub(Kb + i + 1 + (long)j * ld, a[i + 1 + j * DP]);
            else Kb[i + 1 + (long)j * ld] = a[i + 1 + j * DP];
        }
    }
}
// Round 4: L' AS THE FACTOR IS WRITTEN.  The forward sweep of the solves reads L' from the upper triangle (coalesced column
// dots); until round 3 a pass over the whole factor behind the factorisation put it there (k_mirror_lower: 91 us at n = 8192,
// 4 ms of a 64-problem config-5 pass).  Now whoever stores a piece of L stores its transpose too, as 32-byte pieces of 128-byte
// runs: the diagonal kernel for the block's own columns (below), the TRSM strips for theirs (wave_store_T).  Plain stores:
// nobody reads the upper triangle before the solves.
// columns [c, c+16) of the LDS image, rows below the diagonal: K[j, i] = L[i][j] for i > j; `t` of `nt` threads
__device__ __forceinline__ void diag_store_panel_T(const double *a, double *Kb, long ld, int c, int t, int nt) {
#ifdef CIP_NO_TSTORE
    return;                                                  // A/B partner (with CIP_LDLT_MIRROR=1): tools/build_variant.sh notstore diag.hip -DCIP_NO_TSTORE
#endif
    for (int e = t; e < CIP_NB * 4; e += nt) {
        const int i = e >> 2, j0 = c + 4 * (e & 3);             // row i of L, columns j0 .. j0+3 -> K[j0 .. j0+3, i]
        if (i <= j0) continue;
        double *dst = Kb + j0 + (long)i * ld;
        const double v0 = a[i + j0 * DP], v1 = a[i + (j0 + 1) * DP], v2 = a[i + (j0 + 2) * DP], v3 = a[i + (j0 + 3) * DP];
        if (i > j0 + 3) {
            *(v2d *)dst = (v2d){v0, v1};
            *(v2d *)(dst + 2) = (v2d){v2, v3};
        } else {                                                // the row crosses the diagonal inside this piece
            dst[0] = v0;
            if (i > j0 + 1) dst[1] = v1;
            if (i > j0 + 2) dst[2] = v2;
        }
    }
}
// A wave's 16 x 16 tile in the accumulator layout -- lane (l15, g), register q = element (row l15, column 4q + g) -- stored
// TRANSPOSED: dst[c + r * ld] = tile[r][c].  Through a 16 x 17 LDS scratch of the wave's own: lane (r = lane / 4, cq = lane % 4)
// then holds columns 4 cq .. 4 cq + 3 of row r, 32 contiguous bytes of the 128-byte run of that row.
__device__ __forceinline__ void wave_store_T(double *scr, const double (&v)[4], double *dst, long ld, int lane) {
#ifdef CIP_NO_TSTORE
    return;
#endif
    const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) scr[l15 * 17 + 4 * q + g] = v[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int r = lane >> 2, cq = lane & 3;
    const double *sp = scr + r * 17 + 4 * cq;
    const double t0 = sp[0], t1 = sp[1], t2 = sp[2], t3 = sp[3];
    double *d = dst + 4 * cq + (long)r * ld;
    *(v2d *)d = (v2d){t0, t1};
    *(v2d *)(d + 2) = (v2d){t2, t3};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // (the scratch is rewritten by the next call)
    __builtin_amdgcn_wave_barrier();
}
// micro-panel kb's inverse, d and 1/d -> global, write-through (PUB launches publish them per micro-panel)
__device__ __forceinline__ void diag_publish_micro(const double *a, const double *xm, double *xm_out, double *dvec, double *dinv,
                                                   int kb, int t, int nt) {
    for (int i = t; i < 256; i += nt) st_pub(xm_out + kb * 256 + i, xm[kb * 256 + i]);
    if (t < 16) {
        st_pub(dvec + kb * 16 + t, a[128 + (kb * 16 + t) * DP]);
        st_pub(dinv + kb * 16 + t, a[129 + (kb * 16 + t) * DP]);
    }
}

// Factor-only diagonal kernel: L (strictly lower) and d back into K, d / 1/d vectors, and the 8 micro
// inverses Xm (xm_out[kb][k*16 + jj] = inv(L11_kb)[jj][k]) for the TRSM and the block-inverse kernels.
// WAIT: the block is being updated by other workgroups of the SAME launch (the previous panel's in-block update of this
// block's lower triangle: three quarter tiles in k_ldlt_diag_upd, 36 one-per-wave tiles in k_ldlt_panel, written with
// agent-scope stores and counted on `ready`); thread 0 polls the counter up to `ready_target`, one acquire fence, then the
// block is read with agent-scope loads.
// PUB: every micro-panel is published as soon as it is final -- its columns of L, its micro inverse, d and 1/d written
// through by the helper waves under wave 0's next serial step, then one count per helper wave on `stage`: stage >= NH (kb + 1)
// <=> micro-panels 0..kb are readable by the other workgroups of the launch (8 NH at the end).
// NW: waves of the workgroup (all of them call this); wave 0 is the serial one, waves 4, 8, .. idle (they would share its
// SIMD), the others are its NH = NW - NW / 4 helpers.
template <bool WAIT, bool PUB = false, int NW = 4>
__device__ __forceinline__ void diag_body(double *sm, double *Kb, long ld, double *xm_out, double *dvec, double *dinv,
                                          int *info, int col0, PivotSigns sg, const unsigned *ready, unsigned *stage = nullptr,
                                          unsigned ready_target = 3u) {
    double *a = sm;
    double *xm = sm + XM_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;

    // NW > 4: waves 4, 8, .. would share wave 0's SIMD (waves are dealt round-robin to the four SIMDs) and slow the serial
    // wave by 30 % (measured, whatever the priorities): they idle at the barriers, the others are the helpers
    constexpr int NH = NW - NW / 4;
    const bool idle = wave != 0 && (wave & 3) == 0;
    const int hid = wave - 1 - (wave >> 2);                    // helper index 0 .. NH - 1
    __builtin_amdgcn_s_setprio(3);
    PANEL_STAMP(0, WAIT && PUB && tid == 0);
    if (WAIT) {
        if (tid == 0) {
            const long t0 = __builtin_amdgcn_s_memtime();
            while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ready_target) {
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memtime() - t0 > 2000000000L) { atomicCAS(info + 3, 0, -9); break; }   // ~1 s of shader clock: never hang the GPU
            }
            // (round 5) NO agent-scope acquire here: it is a buffer_inv of this CU's vector cache, whose completion the wait below then
            // sat out (~1.7 us, MI355X_MICROARCH.md fence table) -- and nothing in this workgroup reads the block through that cache: the
            // producers stored every byte write-through (st_pub) and drained before counting, and every load below is an sc1 buffer load
            // to registers (the guide's "valid form" without the acquire)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            PANEL_STAMP(4, PUB);
        }
        __syncthreads();
        // 16-byte loads that bypass the non-coherent caches (buffer load with the sc1 policy bit = agent scope)
        typedef int v4i_t __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)Kb, 0, 0x7fffffff, 0x00020000);
        auto put = [&](int e, v4i_t t) {
            const int i = 2 * (e & 63), j = e >> 6;
            v2d v = __builtin_bit_cast(v2d, t);
            if (i < j) v.x = 0.0;                                    // strictly upper part -> 0
            if (i + 1 < j) v.y = 0.0;
            *(v2d *)(a + i + j * DP) = v;
        };
        auto get = [&](int e) -> v4i_t {                              // pair index e: rows 2 (e & 63), + 1; column e >> 6; pairs wholly above the diagonal are not fetched
            const int i = 2 * (e & 63), j = e >> 6;
            if (i + 1 >= j) return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((i + (long)j * ld) * 8), 0, 16);
            return (v4i_t){0, 0, 0, 0};
        };
        if constexpr (NW > 4) {
            // Round 5: the first micro-panel's columns FIRST.  A workgroup fetches a freshly written 72 KB from the other XCDs' side at
            // ~65 GB/s (MI355X_MICROARCH.md, handoff-payload): 2.3 us between "ready" and "block in LDS", all of it in front of A(0).
            // A(0) needs columns 0..15 only: all eight waves fetch those (two loads per thread), one barrier, the serial wave starts,
            // and the other seven waves fetch columns 16..127 beside it (the barrier behind A(0) is in front of their first use).
            {
                v4i_t t0 = get(tid), t1 = get(512 + tid);
                put(tid, t0); put(512 + tid, t1);
            }
            if (tid < 3) *(unsigned *)(a + 130 + tid) = 0u;
            __syncthreads();
            if (wave != 0) {
                constexpr int REST = 64 * (NW - 1), PER = (8192 - 1024 + REST - 1) / REST;
                v4i_t t[PER];
#pragma unroll
                for (int q = 0; q < PER; ++q) { const int e = 1024 + q * REST + (tid - 64); t[q] = e < 8192 ? get(e) : (v4i_t){0, 0, 0, 0}; }
#pragma unroll
                for (int q = 0; q < PER; ++q) { const int e = 1024 + q * REST + (tid - 64); if (e < 8192) put(e, t[q]); }
            }
        } else {
            // every wave of the workgroup takes part, and the pairs that lie wholly above the diagonal are not fetched at all
            constexpr int LT = 256, PER = 8192 / LT;
            if (tid < LT) {
                v4i_t t[PER];
#pragma unroll
                for (int q = 0; q < PER; ++q) t[q] = get(q * LT + tid);
#pragma unroll
                for (int q = 0; q < PER; ++q) put(q * LT + tid, t[q]);
            }
        }
    } else if (NW > 4 && !(DIAG_SKIP & 16)) {
        // (first panel of an outer block, standalone diagonal kernel: plain loads, the same two phases)
        auto put = [&](int e, v2d v) {
            const int i = 2 * (e & 63), j = e >> 6;
            if (i < j) v.x = 0.0;
            if (i + 1 < j) v.y = 0.0;
            *(v2d *)(a + i + j * DP) = v;
        };
        auto get = [&](int e) -> v2d {
            const int i = 2 * (e & 63), j = e >> 6;
            return (i + 1 >= j) ? *(const v2d *)(Kb + i + (long)j * ld) : (v2d){0.0, 0.0};
        };
        {
            v2d t0 = get(tid), t1 = get(512 + tid);
            put(tid, t0); put(512 + tid, t1);
        }
        if (tid < 3) *(unsigned *)(a + 130 + tid) = 0u;
        __syncthreads();
        if (wave != 0) {
            constexpr int REST = 64 * (NW - 1), PER = (8192 - 1024 + REST - 1) / REST;
            v2d t[PER];
#pragma unroll
            for (int q = 0; q < PER; ++q) { const int e = 1024 + q * REST + (tid - 64); t[q] = e < 8192 ? get(e) : (v2d){0.0, 0.0}; }
#pragma unroll
            for (int q = 0; q < PER; ++q) { const int e = 1024 + q * REST + (tid - 64); if (e < 8192) put(e, t[q]); }
        }
    } else if (!(DIAG_SKIP & 16) && (NW == 4 || tid < 256)) diag_load_block(a, Kb, ld, tid);
    if (!(NW > 4 && (WAIT || !(DIAG_SKIP & 16)))) {
        if (tid < 3) *(unsigned *)(a + 130 + tid) = 0u;                 // the phase counts of the loop below (pitch padding of column 0; rows 128 / 129 hold d and 1/d)
        __syncthreads();
    }

    // Schedule per 16-column micro-panel kb (A(0) first):
    //   B(kb)   all waves   : L tiles of the panel rows below (round-robin over the waves)
    //   barrier
    //   wave 0              : trailing update of the NEXT diagonal micro-block, then A(kb+1)      } overlapped
    //   the helper waves    : write-back of micro-panel kb, then the tiles needed next, left-looking }
    //                         (column kb+1 and tile (kb+2, kb+2), steps 0 .. kb at once: diag_tile_left)
    //   barrier
    PANEL_STAMP(2, WAIT && PUB && tid == 0);                            // the block is in LDS
    if (wave == 0) diag_step_a(a, xm, 0, lane, info, col0, sg);
    __syncthreads();
    PANEL_STAMP(6, WAIT && PUB && tid == 0);                            // A(0) done, B(0) starts
#ifndef DIAG_STEP_A_REF
    if constexpr (NW > 4) {
        // Round 5: the micro-panel loop WITHOUT workgroup barriers.  What each wave really waits for:
        //   serial wave, step kb : the helpers' tiles of step kb-1 (its B tile (kb+1, kb) and its C tile (kb+1, kb+1) are among them)
        //   helpers, step kb     : A(kb) (the micro inverse, d, 1/d), the tiles of step kb-1 (their B tiles), then everybody's B(kb)
        // With a barrier behind B and one at the end of the step the serial wave also waited for the helpers' B tiles (which it
        // never reads; two helper waves share a SIMD and an MFMA pipe: 130-590 ticks per step) and for their panel stores (up to
        // 1500 ticks at the step whose four tiles put two tile waves on one SIMD).  Now every working wave counts itself on an LDS
        // word when a phase's LDS writes are done (the DS operations of a wave complete in order) and a reader polls the word it
        // needs; the helpers do their tiles FIRST and write the finished micro-panel back afterwards, beside the next step.
        // tools/diag_bench: 21.2 -> 19.x us per kernel with the other round-5 changes.  The polls are bounded (never hang the GPU).
        unsigned *bflag = (unsigned *)(a + 130), *tflag = (unsigned *)(a + 131), *aflag = (unsigned *)(a + 132);
        // (everything handed over here lives in LDS, and the DS operations of a wave are executed in order: the count's ds_add is
        //  behind the wave's stores without an s_waitcnt -- a workgroup-scope release would be one, ~100 clocks on the serial wave twice
        //  per step -- and a poll that has matched is in front of the loads that follow it.  The fences only pin the compiler's order.)
        auto count = [&](unsigned *f) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (lane == 0) __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        auto wait_for = [&](unsigned *f, unsigned want) {
            if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
                const long t0 = __builtin_amdgcn_s_memtime();
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memtime() - t0 > 2000000000L) { atomicCAS(info + 3, 0, -8); break; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        if (!idle) {
            for (int kb = 0; kb < 7; ++kb) {
                const int c = kb * 16;
                double xa[4], di4[4], d4[4];
                BcOperands bo;
                if (wave == 0) {
                    // the serial wave: the poll for the helpers' tiles of step kb - 1 and EVERY operand of its B + C in ONE LDS round trip
                    // (the DS operations of a wave return in order; its own A(kb) stores are in front of them).  The tiles are normally
                    // long done -- if not, wait and load the two tile operands again.
                    const unsigned tf = __hip_atomic_load(tflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    // (round 6, advisor) the operand loads below are PLAIN LDS loads and the acquire fence comes behind them: the hardware
                    // returns a wave's DS operations in order, but nothing kept the COMPILER from hoisting them above the relaxed poll --
                    // a compiler barrier (no instruction) pins the order the one-round-trip form relies on
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        xa[s] = xm[kb * 256 + (g + 4 * s) * 16 + l15];
                        di4[s] = a[129 + (c + g + 4 * s) * DP];
                        d4[s] = a[128 + (c + g + 4 * s) * DP];
                    }
                    bo = diag_step_bc_load(a, kb, l15, g);
                    if (tf < (unsigned)(kb * NH)) {
                        wait_for(tflag, (unsigned)(kb * NH));
                        bo = diag_step_bc_load(a, kb, l15, g);
                    } else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                } else {
                    wait_for(aflag, (unsigned)kb);                             // A(kb): counted by the serial wave (A(0): the barrier above)
                    wait_for(tflag, (unsigned)(kb * NH));                      // the helpers' tiles of step kb - 1
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        xa[s] = xm[kb * 256 + (g + 4 * s) * 16 + l15];       // Aop[jj = l15][k = g + 4s]
                        di4[s] = a[129 + (c + g + 4 * s) * DP];
                        d4[s] = a[128 + (c + g + 4 * s) * DP];
                    }
                }
                DIAG_STAMP(kb, 0, tid == 0);                                  // wave 0: starts B(kb)
                DIAG_STAMP(kb, 4, tid == DIAG_TIMING_TID);
                v4d U1 = (v4d){0.0, 0.0, 0.0, 0.0};
                if (!(DIAG_SKIP & 4)) {
                    // (kb+1, kb+1): steps 0 .. kb-1 were applied by a helper during step kb-1, step kb is this wave's
                    if (wave == 0) U1 = diag_step_bc_perm(a, kb, l15, g, bo, xa, di4, d4);
                    else { for (int it = kb + 2 + hid; it < 8; it += NH) diag_step_b(a, it, c, l15, g, xa, di4); }
                }
                DIAG_STAMP(kb, 1, tid == 0);                                  // B (serial wave: B + C) done
                DIAG_STAMP(kb, 5, tid == DIAG_TIMING_TID);
                count(bflag);
                DIAG_STAMP(kb, 2, tid == 0);
                if (PUB && wave != 0 && kb > 0) {
                    // the stage count of micro-panel kb - 1, whose write-through stores this wave issued at the end of the last step:
                    // they have had the serial wave's A(kb) and this B to land (waiting for them right behind the stores put a store
                    // round trip -- 1.5-2 us inside a busy launch -- on every helper's step, and the serial wave then waited for the
                    // helpers' tiles)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) atomicAdd(stage, 1u);
                }
                if (wave == 0) {
                    if (!(DIAG_SKIP & 4)) diag_step_a<true>(a, xm, kb + 1, lane, info, col0, sg, U1);
                    else diag_step_a(a, xm, kb + 1, lane, info, col0, sg);
                    count(aflag);
                    PANEL_STAMP(20 + kb, WAIT && PUB && tid == 0);              // A(kb + 1) done
                    if (PUB && kb == 6) {
                        // the LAST stage of the launch's TRSM strips needs the last micro inverse and 1/d only (no column of L lies below
                        // the last diagonal tile): the serial wave publishes them itself the moment A(7) is done and counts the stage for
                        // all the helpers -- not behind a workgroup barrier, the write-back of the last tile by all waves and their drain
                        // (the strips' last stage is the tail of every panel launch)
#pragma unroll
                        for (int q = 0; q < 4; ++q) st_pub(xm_out + 7 * 256 + lane + 64 * q, xm[7 * 256 + lane + 64 * q]);
                        if (lane < 16) {
                            st_pub(dvec + 112 + lane, a[128 + (112 + lane) * DP]);
                            st_pub(dinv + 112 + lane, a[129 + (112 + lane) * DP]);
                        }
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        // (round 6: in the HIGH half of the word.  Until then it was NH more on the one running count -- and the strips' stage
                        //  6, "count >= 7 NH", came true with the helpers' micro-panel 6 still on its way whenever their write-through stores
                        //  took longer than A(7) and this publication: about one factorisation in 100 000 at order 4096 came out with a few
                        //  64-row strips of one panel computed from the previous contents of column block 6 -- see PANEL_STAGE_LAST)
                        if (lane == 0) atomicAdd(stage, PANEL_STAGE_LAST);
                        PANEL_STAMP(8, WAIT && tid == 0);
                    }
                    DIAG_STAMP(kb, 3, tid == 0);                              // C11 + A(kb+1) done
                } else if (!(DIAG_SKIP & 4)) {
                    wait_for(bflag, (unsigned)((kb + 1) * (NH + 1)));         // everybody's B(kb): the tiles below read them
                    DIAG_STAMP(kb, 6, tid == DIAG_TIMING_TID);                // helper: B of all waves seen
                    // left-looking: the tiles that are needed NEXT -- column kb+1 below its diagonal tile (the panel of step kb+1) and
                    // the diagonal tile (kb+2, kb+2) (wave 0's C11 of step kb+1) -- receive all their steps 0 .. kb now
                    const int ncol = 6 - kb;
                    const int ntl = ncol + (kb + 2 <= 7 ? 1 : 0);
                    for (int t = hid; t < ntl; t += NH) {
                        const int it = t < ncol ? kb + 2 + t : kb + 2, jt = t < ncol ? kb + 1 : kb + 2;
                        diag_tile_left(a, it, jt, kb + 1, l15, g);
                    }
                    count(tflag);
                    DIAG_STAMP(kb, 7, tid == DIAG_TIMING_TID);                // helper: tiles done and counted
                    // micro-panel kb is final: written back (and published) now, beside the serial wave's step and the next B
#ifdef DIAG_DEBUG_SLOW_HELPERS
                    // TEST BUILD (tools/stage_mix_demo.sh, tests/test_gpu_panel_chain.py): the helpers' write-back of micro-panel 6 comes
                    // DIAG_DEBUG_SLOW_HELPERS ticks of s_memtime late (2000 were enough to turn the old counting over), as a slow store
                    // would make it -- the factor must not change
                    if (PUB && kb == 6) { const long t0 = __builtin_amdgcn_s_memtime(); while (__builtin_amdgcn_s_memtime() - t0 < (long)(DIAG_DEBUG_SLOW_HELPERS)) __builtin_amdgcn_s_sleep(8); }
#endif
                    if (!(DIAG_SKIP & 16)) diag_store_panel<PUB>(a, Kb, ld, c, hid * 64 + lane, 64 * NH);
                    if (PUB) diag_publish_micro(a, xm, xm_out, dvec, dinv, kb, hid * 64 + lane, 64 * NH);      // (counted in the next step)
                } else count(tflag);
            }
            if (PUB && wave != 0) {                                            // micro-panel 6's count
                // (round 6) not before EVERY helper has issued its count of micro-panel 5: inside the loop a wave cannot run a step
                // ahead of another one's count (it waits for everybody's tiles of the step before, counted behind that count), here
                // nothing held it -- a helper whose stores of micro-panel 5 were slow could be overtaken, and the running count said
                // "6 NH" with that micro-panel incomplete.  Every helper counts `tflag` in step 6 behind its count of micro-panel 5.
#ifndef DIAG_OLD_STAGE_COUNT
                wait_for(tflag, 7u * NH);
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(stage, 1u);
            }
        }
        __syncthreads();                                                       // A(7) and the helpers' last stores
    } else
#endif
    {
    for (int kb = 0; kb < 7; ++kb) {
        const int c = kb * 16;
        double xa[4], di4[4], d4[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            xa[s] = xm[kb * 256 + (g + 4 * s) * 16 + l15];       // Aop[jj = l15][k = g + 4s]
            di4[s] = a[129 + (c + g + 4 * s) * DP];
            d4[s] = a[128 + (c + g + 4 * s) * DP];
        }
        DIAG_STAMP(kb, 0, tid == 0);                                  // wave 0: past the barrier that follows A(kb)
        DIAG_STAMP(kb, 4, tid == DIAG_TIMING_TID);
        if (!(DIAG_SKIP & 4))
        {
            if (NW == 4) { for (int it = kb + 1 + wave; it < 8; it += 4) diag_step_b(a, it, c, l15, g, xa, di4); }
            else if (wave == 0) diag_step_b(a, kb + 1, c, l15, g, xa, di4);
            else if (!idle) { for (int it = kb + 2 + hid; it < 8; it += NH) diag_step_b(a, it, c, l15, g, xa, di4); }
        }
        DIAG_STAMP(kb, 1, tid == 0);                                  // B done
        DIAG_STAMP(kb, 5, tid == DIAG_TIMING_TID);
        __syncthreads();
        DIAG_STAMP(kb, 2, tid == 0);                                  // past the B barrier
        if (wave == 0) {
            // (kb+1, kb+1): steps 0 .. kb-1 were applied by a helper during step kb-1, step kb is this wave's
#ifdef DIAG_STEP_A_REF
            if (!(DIAG_SKIP & 4)) diag_step_c(a, kb + 1, kb + 1, c, l15, g, d4);
            diag_step_a(a, xm, kb + 1, lane, info, col0, sg);
#else
            if (!(DIAG_SKIP & 4)) diag_step_a<true>(a, xm, kb + 1, lane, info, col0, sg, diag_step_c_perm(a, kb, l15, g, d4));
            else diag_step_a(a, xm, kb + 1, lane, info, col0, sg);
#endif
            DIAG_STAMP(kb, 3, tid == 0);                              // C11 + A(kb+1) done
            PANEL_STAMP(20 + kb, WAIT && PUB && tid == 0);
        } else if (!(DIAG_SKIP & 4) && !idle) {
            // micro-panel kb is final (A(kb) and B(kb) are behind the barrier): the helper waves write it back now, under
            // wave 0's serial step, instead of in a store phase at the end of the kernel
            // (round 5) from micro-panel 3 on two or more helpers have no tile below: they alone write the panel back, the tile
            // waves go straight to their tiles
            const int ntile_waves = 7 - kb < NH ? 7 - kb : NH;           // = min(ntl, NH), ntl as computed below
            const int nfree = NH - ntile_waves;
            const bool split = NW > 4 && nfree >= 2;
            const bool storer = !split || hid >= ntile_waves;
            const int st_t = split ? (hid - ntile_waves) * 64 + lane : hid * 64 + lane, st_n = split ? 64 * nfree : 64 * NH;
            if (storer) {
                if (!(DIAG_SKIP & 16)) diag_store_panel<PUB>(a, Kb, ld, c, st_t, st_n);
                if (PUB) diag_publish_micro(a, xm, xm_out, dvec, dinv, kb, st_t, st_n);
            }
            DIAG_STAMP(kb, 6, tid == DIAG_TIMING_TID);                // helper: panel stores issued
#ifndef DIAG_STEP_A_REF
            // left-looking: the tiles that are needed NEXT -- column kb+1 below its diagonal tile (the panel of step kb+1) and
            // the diagonal tile (kb+2, kb+2) (wave 0's C11 of step kb+1) -- receive all their steps 0 .. kb now
            const int ncol = 6 - kb;
            const int ntl = ncol + (kb + 2 <= 7 ? 1 : 0);
            for (int t = hid; t < ntl; t += NH) {
                const int it = t < ncol ? kb + 2 + t : kb + 2, jt = t < ncol ? kb + 1 : kb + 2;
                diag_tile_left(a, it, jt, kb + 1, l15, g);
            }
#else
            // round 2: step kb on every tile of rows kb+2.. ((kb+1, kb+1) is wave 0's), dealt round-robin to the helper waves, three at a time
            const int ntile = (6 - kb) * (9 - kb) / 2;
            for (int idx = hid; idx < ntile; idx += 3 * NH) {
                const int left = (ntile - idx + NH - 1) / NH;      // tiles idx, idx + NH, idx + 2 NH that exist
                if (left >= 3) diag_step_c_multi<3>(a, kb, idx, NH, c, l15, g, d4);
                else if (left == 2) diag_step_c_multi<2>(a, kb, idx, NH, c, l15, g, d4);
                else diag_step_c_multi<1>(a, kb, idx, NH, c, l15, g, d4);
            }
#endif
            #ifndef DIAG_TIMING_END
            DIAG_STAMP(kb, 7, tid == DIAG_TIMING_TID);                // helper: C tiles done
#endif
            if (PUB) {                                            // the wave's stores have landed -> its count
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(stage, 1u);
            }
            // L' of the micro-panel: NOT here (round 5).  Nobody reads the upper triangle before the solves, and the helper waves
            // -- two per SIMD, the serial wave alone on the fourth -- are what the serial wave waits for at this barrier from
            // micro-panel 2 on (tools/diag_bench -DDIAG_TIMING -DDIAG_TIMING_END -DDIAG_TIMING_TID=..: a helper with a tile on a SIMD
            // that hosts two tile waves needs 5100 ticks for stores + tile + transposed stores beside the serial wave's 3900).
            // (The 128 x 128 block's own transpose is not written at all since round 5: the solves read L' from the upper triangle
            // only OUTSIDE their Bs-wide diagonal blocks.  The upper triangle of the diagonal blocks is UNDEFINED after a factorisation.)
#ifdef DIAG_TIMING_END
            DIAG_STAMP(kb, 7, tid == DIAG_TIMING_TID);                // (instead of "tiles done": the helper's whole phase, transposed stores included)
#endif
        }
        __syncthreads();
    }
    }
    PANEL_STAMP(7, WAIT && PUB && tid == 0);                            // last pivot done
    if (PUB && NW > 4) {
#ifndef DIAG_STEP_A_REF
        diag_store_panel<true>(a, Kb, ld, 112, tid, 64 * NW);              // the last diagonal tile (the stage was counted by the serial wave in the loop)
        return;
#endif
    }
    if (PUB) {
        diag_store_panel<true>(a, Kb, ld, 112, tid, 64 * NW);
        diag_publish_micro(a, xm, xm_out, dvec, dinv, 7, tid, 64 * NW);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) atomicAdd(stage, PANEL_STAGE_LAST);
        PANEL_STAMP(8, WAIT && tid == 0);
        // (round 5: the diagonal block's own transpose is NOT written any more.  The solves read L' from the upper triangle only
        //  OUTSIDE their Bs-wide diagonal blocks (ldlt.hip: cip_ldlt_solve, K + C0 + (C0 + Bs) ld; Bs >= 128), the diagonal blocks
        //  go through the explicit block inverses: these 128 x 128 transposes -- 1.3 us at the tail of every panel launch since
        //  round 4 -- had no reader.  The strips still store theirs: wave_store_T.)
        return;
    }

    // ---- last micro-panel, d and the micro inverses out; the strictly upper part of K is left untouched
    if (!(DIAG_SKIP & 16)) diag_store_panel(a, Kb, ld, 112, tid, 64 * NW);
    if (tid < CIP_NB) {
        dvec[tid] = a[128 + tid * DP];
        dinv[tid] = a[129 + tid * DP];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (NW == 4 || tid < 256) xm_out[q * 256 + tid] = xm[q * 256 + tid];
}

// NW waves: 8 by default (CIP_DIAG_WAVES = 4 / 8 / 12).  The serial wave was waiting for its three helpers at the first
// micro-panels (tools/diag_bench, per-phase clocks: at kb = 0 the helpers' panel stores + 9 trailing tiles each take twice
// wave 0's C + A); with six or nine helpers it no longer does: 29.3 -> 24.0 / 23.4 us per launch.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_ldlt_diag128_v2(double *Kb, long ld, double *xm_out, double *dvec, double *dinv,
                                                             int *info, int col0, PivotSigns sg, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO5(cb, Kb, xm_out, dvec, dinv, info);
    extern __shared__ __attribute__((aligned(16))) double sm[];
    diag_body<false, false, NW>(sm, Kb, ld, xm_out, dvec, dinv, info, col0, sg, nullptr);
}

// The diagonal kernel of inner panel t >= 1 FUSED with the in-block update of panel t-1: one launch instead of two on the
// serial chain (diag -> TRSM -> update -> diag became [update tiles || diag] -> TRSM):
//   workgroup 0        waits for the three quarter tiles of the update that are its block's lower triangle, then its LDL'
//   workgroups 1..3    those three tiles, C read and written with agent-scope accesses (gemm_tile_64<.., SC1C>), then
//                      one atomicAdd on the launch's `ready` counter
//   the others         the remaining 64x64 tiles of C -= W L'.  The launch's 160 KB per workgroup limits them to one per CU;
//                      they use it: all of K = 128 staged at once (gemm_tile_64_k128: one global round trip per tile)
// WAIT = false is the timing experiment CIP_FUSE_DIAG=2 (workgroup 0 does not wait: wrong results).
template <bool WAIT>
__global__ __launch_bounds__(256) void k_ldlt_diag_upd(double *Kb, long ld, double *xm_out, double *dvec, double *dinv,
                                                        int *info, int col0, PivotSigns sg, unsigned *ready, GemmArgs g,
                                                        CipBatch cb) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (blockIdx.x == 0) {
        CIP_BATCH_GUARD(cb);
        CIP_BO6(cb, Kb, xm_out, dvec, dinv, info, ready);
        diag_body<WAIT>(sm, Kb, ld, xm_out, dvec, dinv, info, col0, sg, ready);
        return;
    }
    bool live;
    (void)gemm_batch_prologue(g, cb, live);
    if (!live) return;
    ready = cip_bo(ready, cb);
    __builtin_amdgcn_s_setprio(3);
    const int tm = g.M / SB;
    const int b = (int)blockIdx.x;
    if (b <= 3) {
        // (0,0), (64,0), (64,64): the block's lower triangle, dispatched first
        const long i0 = (b == 1) ? 0 : SB, j0 = (b == 3) ? SB : 0;
        gemm_tile_64_k128<true>(g, sm, i0, j0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(ready, 1u);
        return;
    }
    // the other tiles in column-major order, skipping t = 0, 1, tm, tm + 1 (the block's four quarter tiles; the
    // strictly-upper one, t = tm, is never referenced)
    const int tq = b - 4;
    const int t = (tq < tm - 2) ? tq + 2 : tq + 4;
    gemm_tile_64_k128<false>(g, sm, (long)(t % tm) * SB, (long)(t / tm) * SB);
}

// ---------------------------------------------------------------------------------------------------------------------
// One launch per panel: the diagonal kernel, the previous panel's in-block update AND this panel's TRSM, the TRSM running
// BEHIND the diagonal kernel micro-panel by micro-panel instead of after it (k_trsm_subst needs ~11.5 us after the last
// pivot: a launch, one memory round trip and 144 dependent MFMAs per wave; here all but the last 4 MFMAs of every wave are
// done when the last pivot is):
//   workgroup 0          diag_body<WAIT, PUB>: publishes each 16-column micro-panel (L columns, micro inverse, 1/d) as
//                        soon as it is final and counts on `stage`
//   workgroups 1..9      (UPD) the in-block update of the block itself, which workgroup 0 waits for: one 16x16 tile per wave
//                        (diag_block_producer)
//   the next `strips`    one 64-row strip of the rows below the block: (UPD) its two update tiles of this panel's columns,
//                        then the substitution of k_trsm_subst with the operands of stage kb fetched (agent-scope loads)
//                        once stage >= PANEL_NH (kb + 1):  W[:,kb] = T_kb inv(L11[kb][kb])',  T_{kb+1} = A21[:,kb+1] - sum W[:,q] L11[kb+1][q]'
//   the others           (UPD) the remaining tiles of the in-block update
// The same operations on the same operands in the same order as the separate launches: bit-identical factors.
// Who waits for whom inside the launch: the strips wait for workgroup 0 (lower index: dispatched earlier, so resident or
// done); workgroup 0 waits for the producers 1..9 -- HIGHER indices, dispatched right behind it.  Workgroups are dispatched
// in index order (x, then z = the problem of a lock-step group): the only thing that can stand between a workgroup 0 and its
// producers is a chip full of EARLIER workgroups, all of which can finish without them (the first problem's group is always
// complete on the chip), so every wait ends; with the chain alone on the handle's stream the ten are resident at once and
// the wait is the producers' own time.  Lock-step groups of up to ~4 rounds of the chip take this launch too (ldlt.hip:
// factor_outer_panels).  Every wait is bounded (~1 s of shader clock) and then raises info[3] instead of hanging.
#ifndef PANEL_WAVES
#define PANEL_WAVES 8                   // waves of a k_ldlt_panel workgroup: the diagonal kernel's; the other roles use the first four
#endif
#define PANEL_NH (PANEL_WAVES - PANEL_WAVES / 4)      // helper waves of the diagonal kernel = counts on `stage` per published micro-panel
struct TrsmStrips {
    double *Ap; long ld;              // rows below the diagonal block, this panel's 128 columns
    const double *L11;                // the diagonal block (leading dimension ld)
    const double *xm, *dinv;          // micro inverses / 1/d of the block (published by workgroup 0)
    double *W; long ldw;
    int strips;                       // rows / 64
    int pair;                         // round 5, k_ldlt_panel<true> in lock-step groups: a strip workgroup carries TWO strips (see the kernel)
    int nprod;                        // producer workgroups of k_ldlt_panel<true>: 36 (one wave each; one problem) or 9 (four waves each; lock-step groups)
};
// The stage word of a panel launch: the LOW half counts the helper waves' publications of micro-panels 0 .. 6 (PANEL_NH each, in
// order: low >= PANEL_NH (kb + 1) <=> micro-panels 0 .. kb are readable), the HIGH half is set by the serial wave when micro-panel 7's
// inverse and 1/d are (no column of L lies below the last diagonal tile, and the serial wave publishes them itself, ahead of the
// helpers' last stores).  Two fields because the two kinds of count are not ordered against each other.
__device__ __forceinline__ bool stage_reached(unsigned x, int kb) {
#ifdef DIAG_OLD_STAGE_COUNT
    return x >= (unsigned)PANEL_NH * (unsigned)(kb + 1);
#endif
    return kb < 7 ? (x & 0xffffu) >= (unsigned)PANEL_NH * (unsigned)(kb + 1) : (x >> 16) != 0u;
}
__device__ __forceinline__ unsigned strip_wait(unsigned v, int kb, const unsigned *stage, unsigned *slot, int *info) {
    if (stage_reached(v, kb)) return v;
    if (threadIdx.x == 0) {
        const long t0 = __builtin_amdgcn_s_memtime();
        unsigned x;
        while (!stage_reached(x = __hip_atomic_load(stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), kb)) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memtime() - t0 > 2000000000L) { atomicCAS(info + 3, 0, -9); x = 0xffffffffu; break; }   // never hang the GPU
        }
        *slot = x;
    }
    __syncthreads();
    const unsigned r = *slot;
    __syncthreads();
    return r;
}
// Round 3: RIGHT-looking inside the wave.  The wave keeps the eight 16-column blocks T_0 .. T_7 of its rows in registers; at
// stage kb it finishes W[:,kb] = T_kb inv(L11[kb][kb])' (4 MFMAs) and applies it to ALL later blocks at once,
// T_r -= W[:,kb] L11[r][kb]' for r > kb: 7 - kb independent chains of 4 MFMAs that pipeline.  Round 2 built T_{kb+1} when it
// was needed, from all earlier W blocks: a chain of 4 (kb + 1) DEPENDENT MFMAs (~100 clocks each), 28 of them after the
// diagonal kernel's last micro-panel -- the tail of every panel launch.  Per block the same MFMAs on the same operands in
// the same order (qq ascending): bit-identical.
// the wait of a strip whose four waves share the workgroup with a tile group (k_ldlt_panel<true>): no barrier, every wave
// polls for itself (one blocking agent-scope load per round trip: four pollers per strip do not load the L2 channel)
__device__ __forceinline__ unsigned strip_wait_wave(unsigned v, int kb, const unsigned *stage, int *info) {
    if (stage_reached(v, kb)) return v;
    unsigned x = 0u;
    if ((threadIdx.x & 63) == 0) {
        const long t0 = __builtin_amdgcn_s_memtime();
        while (!stage_reached(x = __hip_atomic_load(stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), kb)) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memtime() - t0 > 2000000000L) { atomicCAS(info + 3, 0, -9); x = 0xffffffffu; break; }   // never hang the GPU
        }
    }
    return (unsigned)__builtin_amdgcn_readfirstlane((int)x);
}
// tscr: 4 x 272 doubles of LDS (one transposition scratch per wave) for the L' stores
template <bool WAVEWAIT>
__device__ __forceinline__ void trsm_strip_pipelined(const TrsmStrips &tr, int strip, const unsigned *stage, unsigned *slot, int *info, double *tscr) {
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const long row = (long)strip * 64 + wave * 16 + l15;
    double *scr = tscr + wave * 272;
    // L' of this wave's rows: K[c0 + k, c0 + 128 + row0 + r], row0 = 64 strip + 16 wave (L11 = the diagonal block at (c0, c0))
    double *ut = const_cast<double *>(tr.L11) + (long)(CIP_NB + strip * 64 + wave * 16) * tr.ld;
    double *ap = tr.Ap + row + (long)g * tr.ld;
    double *wp = tr.W + row + (long)g * tr.ldw;
    v4d T[8];
    unsigned v = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) T[r][q] = ap[(long)(r * 16 + 4 * q) * tr.ld];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        v = WAVEWAIT ? strip_wait_wave(v, kb, stage, info) : strip_wait(v, kb, stage, slot, info);
        // one batch of loads per stage: the micro inverse and 1/d of kb and column block kb of L11 below its diagonal tile
        double xo[4], dv[4], lo[7][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            xo[s] = ld_pub(tr.xm + kb * 256 + (g + 4 * s) * 16 + l15);
            dv[s] = ld_pub(tr.dinv + kb * 16 + 4 * s + g);
        }
#pragma unroll
        for (int r = 1; r < 8; ++r)
            if (r > kb) {
#pragma unroll
                for (int s = 0; s < 4; ++s) lo[r - 1][s] = ld_pub(tr.L11 + r * 16 + l15 + (long)(kb * 16 + 4 * s + g) * tr.ld);
            }
        v4d w = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) w = MFMA(xo[s], T[kb][s], w);
        double wneg[4], lq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long col = kb * 16 + 4 * q;
            wneg[q] = -w[q];
            lq[q] = w[q] * dv[q];
            wp[col * tr.ldw] = w[q];
            ap[col * tr.ld] = lq[q];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int r = 1; r < 8; ++r)
                if (r > kb) T[r] = MFMA(lo[r - 1][s], wneg[s], T[r]);
        wave_store_T(scr, lq, ut + kb * 16, tr.ld, lane);          // behind the MFMAs' issue: off the chain
    }
}
// One 16x16 tile (it >= jt) of the diagonal block's update C -= W L' (K = 128) per WAVE, operands straight from L2 into
// registers, 32 dependent MFMAs: the producer side of workgroup 0's wait in k_ldlt_panel, 36 waves in 9 workgroups instead
// of three 64x64 tiles (3.4 us of MFMAs per wave behind an LDS staging pass).  Same k order from a zero accumulator and the
// same C + alpha acc as gemm_tile_64_k128: identical bits (the 16x16 tiles above the diagonal, which nobody reads, are
// left alone).  Written through; one count per wave on `ready` (36 = complete).
__device__ __forceinline__ void diag_block_producer(const GemmArgs &g, int tile, unsigned *ready, int col0 = 0) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    int it = 0;
    while ((it + 1) * (it + 2) / 2 <= tile) ++it;              // row-major over the lower triangle of the 8 x 8 tile grid
    const int jt = tile - it * (it + 1) / 2;
    const double *ap = g.A + it * 16 + l15 + (long)l4 * g.lda;   // W rows of the tile's rows
    const double *bp = g.B + jt * 16 + l15 + (long)l4 * g.ldb;   // L rows of the tile's columns
    double *cp = g.C + (it * 16 + l15) + (long)(jt * 16 + l4) * g.ldc;
    double wa[32], lb[32], cpre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cpre[q] = cp[(long)(4 * q) * g.ldc];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
        wa[ks] = ap[(long)(4 * ks) * g.lda];
        lb[ks] = bp[(long)(4 * ks) * g.ldb];
    }
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#ifdef PANEL_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PANEL_STAMP(10, tile == 0 && lane == 0);                       // (timing build) producer of tile (0, 0): operands in registers
#endif
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) acc = MFMA(lb[ks], wa[ks], acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) st_pub(cp + (long)(4 * q) * g.ldc, cpre[q] + g.alpha * acc[q]);
    PANEL_STAMP(11, tile == 0 && lane == 0);                       // MFMAs done, stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PANEL_STAMP(21 + 7, tile == 0 && lane == 0);                   // stores landed (slot 28)
    if (lane == 0) atomicAdd(ready, 1u);
}
// 36 tiles, one per wave.  Round 5: ONE wave per producer workgroup (36 workgroups), not four (9).  A producer wave fetches 32 KB of
// operands (its tile's 16 rows of W and of L, K = 128) that the previous launch's strips wrote on other XCDs, and a workgroup fetches
// such bytes at ~65 GB/s whatever its wave count (MI355X_MICROARCH.md, handoff-payload): with four waves the operands were in the
// registers 2.7-3.2 us after the launch's start (tools/panel_stamps.py), the 32 MFMAs take 1.0.  The producers are gone after ~3 us
// and the tile workers behind them in the grid take their CUs (the launcher still budgets PANEL_PRODUCER_CUS for them: the grid may
// exceed the chip by the difference -- only workers, who wait for nobody, are dispatched late).  -DPANEL_PRODUCERS=9 restores.
// Lock-step groups keep nine producer workgroups of four waves per problem: B x 36 workgroups in front of the strips cost more than
// the faster fetch buys (8 / 16 problems of order 2048: 15.2 / 23.9 ms per pass with 9, 15.4 / 25.5 with 36).
#ifndef PANEL_PRODUCERS
#define PANEL_PRODUCERS 36
#endif
#define PANEL_PRODUCER_CUS 9
// Round 3, the update tiles as a QUEUE.  The launch's LDS size is the diagonal kernel's (160 KB), so every workgroup has a
// CU to itself -- and at the top of the matrix 126 strips held 126 CUs for the whole launch although they mostly wait, while
// the 1000 update tiles of the block's second panel queued for the other 120 (56 / 47 / 39-us launches where the chain needs
// 28).  Now every workgroup behind the producers is TWO groups of four waves (64 KB of LDS and an LDS meeting word each,
// gemm_tile_64_k128_grp): group 0 of the first `strips` workgroups is the strip (its two update tiles, then the TRSM,
// every wave polling `stage` for itself), every other group -- the strips' second halves, both halves of the `workers`
// workgroups behind them, and the strips' own waves once their TRSM is done -- draws 64x64 tiles from `tileq` until it runs
// dry.  Nobody waits for a tile job, a tile job waits for nobody: forward progress as before.  Each tile is computed by
// one group in the order of gemm_tile_64_k128: the factor's bits do not depend on who drew what.
#define PANEL_GRP_DOUBLES 8192          // 64 KB per group; the control words (two meeting counters, two queue slots, a flag) behind both
#define PANEL_NONE 0xffffffffu
__device__ __forceinline__ void panel_tile_jobs(const GemmArgs &g, double *lds, int gt, GrpBar &bar, unsigned *tileq, unsigned *slot,
                                                unsigned first_tile, unsigned nstatic) {
    // column-major from tile column 2 on, the tiles on and below the diagonal only (C's first row is its first column's:
    // tile (i, j) with i < j lies in the upper triangle, which nobody reads).  Column-major on purpose: the tiles in flight
    // at any time are neighbours in a column -- contiguous 512-byte pieces of the same 64 columns of C, spread over the
    // memory channels; a row-major order (tried with one queue per XCD, so that the eight tiles of a tile row would share
    // their W rows in one L2) puts them 64 KB x 64 apart on the same channels and ran 35-70 % slower.
    const int tm = g.M / SB, tn = g.N / SB;
    const unsigned ntiles = (unsigned)(tm * (tn - 2) - (tn * (tn - 1) / 2 - 1));
    unsigned t = first_tile;                                      // a worker's first job needs no draw; the queue hands out nstatic, nstatic + 1, ..
    if (t == PANEL_NONE) {                                        // late starters (the strips' groups) draw their first one too
        if (gt == 0) *(volatile unsigned *)slot = __hip_atomic_fetch_add(tileq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        grp_barrier(bar);
        t = nstatic + *(volatile unsigned *)slot;
    }
    while (t < ntiles) {
        // index -> tile.  Column pairs (j, j + 1), j = 2, 4, ..: inside a pair the rows i > j, each as (i, j), (i, j + 1) -- two
        // consecutive indices share their 64 KB of W rows, and the two groups of a worker workgroup start on such a couple
        // (2 w, 2 w + 1): one fetch through the fabric instead of two in the first round, when everybody loads at once; behind
        // all pairs the diagonal tiles (j, j), which have no partner (their right neighbour lies above the diagonal)
        const int npair = (tn - 2) / 2;
        int r = (int)t, ti = 0, tj = 0, p = 0;
        for (; p < npair; ++p) {
            const int j = 2 + 2 * p, n = 2 * (tm - j - 1);
            if (r < n) { ti = j + 1 + (r >> 1); tj = j + (r & 1); break; }
            r -= n;
        }
        if (p == npair) { tj = 2 + 2 * r; ti = tj; }               // r-th diagonal tile of an even column
        t = nstatic + gemm_tile_64_k128_grp<true>(g, lds, (long)ti * SB, (long)tj * SB, gt, bar, tileq, slot);
    }
}
template <bool UPD>
__global__ __launch_bounds__(64 * PANEL_WAVES) void k_ldlt_panel(double *Kb, long ld, double *xm_out, double *dvec, double *dinv,
                                                                  int *info, int col0, PivotSigns sg, unsigned *ready, unsigned *stage,
                                                                  unsigned *tileq, GemmArgs g, TrsmStrips tr, int bt, CipBatch cb) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    // small lock-step groups (ldlt.hip: factor_outer_panels, up to 12 problems).  bt == 0: the problem is blockIdx.z -- problem z's
    // workgroups follow those of the problems before it in dispatch order.  bt = B > 0 (round 4): the problem index runs FASTEST,
    // workgroup (role b, problem z) = blockIdx.x = b B + z: the diagonal kernels of all problems are dispatched first, then the
    // producers of all problems, ...; in z-major order problem 7's diagonal kernel sat behind the ~280 workgroups of problems 0..6.
    // Either way every wait is for a workgroup dispatched earlier (role b' < b of the same problem).
    const unsigned bz = bt > 0 ? blockIdx.x % (unsigned)bt : blockIdx.z;
    if (!((cb.mask >> bz) & 1ull)) return;
    {
        const long off = (long)bz * cb.stride;
        Kb = (double *)((char *)Kb + off); xm_out = (double *)((char *)xm_out + off); dvec = (double *)((char *)dvec + off);
        dinv = (double *)((char *)dinv + off); info = (int *)((char *)info + off); ready = (unsigned *)((char *)ready + off);
        stage = (unsigned *)((char *)stage + off); if (tileq) tileq = (unsigned *)((char *)tileq + off);
        g.A = (const double *)((const char *)g.A + off); g.B = (const double *)((const char *)g.B + off); g.C = (double *)((char *)g.C + off);
        tr.Ap = (double *)((char *)tr.Ap + off); tr.L11 = (const double *)((const char *)tr.L11 + off);
        tr.xm = (const double *)((const char *)tr.xm + off); tr.dinv = (const double *)((const char *)tr.dinv + off);
        tr.W = (double *)((char *)tr.W + off);
    }
    const int b = bt > 0 ? (int)(blockIdx.x / (unsigned)bt) : (int)blockIdx.x;
    const int nblocks = bt > 0 ? (int)(gridDim.x / (unsigned)bt) : (int)gridDim.x;
    if (b == 0) {
        diag_body<UPD, true, PANEL_WAVES>(sm, Kb, ld, xm_out, dvec, dinv, info, col0, sg, ready, stage, 36u);
        return;
    }
    if (!UPD) {                                         // first panel of an outer block: strips only, four-wave jobs
        __builtin_amdgcn_s_setprio(3);
        if (threadIdx.x >= 256) return;
        trsm_strip_pipelined<false>(tr, b - 1, stage, (unsigned *)sm, info, sm + 64);
        return;
    }
    if (b <= tr.nprod) {
        __builtin_amdgcn_s_setprio(3);
        PANEL_STAMP(9, threadIdx.x == 0 && b == 1);
        const int pw = 36 / tr.nprod;                       // waves per producer workgroup
        if ((int)threadIdx.x < 64 * pw) diag_block_producer(g, (b - 1) * pw + (int)(threadIdx.x >> 6), ready, col0);
        return;
    }
    const int first = 1 + tr.nprod;
    const int grp = (int)(threadIdx.x >> 8), gt = (int)(threadIdx.x & 255);
    unsigned *ctl = (unsigned *)(sm + 2 * PANEL_GRP_DOUBLES);
    if (threadIdx.x < 5) ctl[threadIdx.x] = 0u;
    __syncthreads();                                    // the one hardware barrier of these workgroups: all eight waves are still here
    GrpBar bar = {ctl + grp, 0u};
    double *lds = sm + grp * PANEL_GRP_DOUBLES;
    // the two halves of worker workgroup w start at once on tiles 2 w and 2 w + 1; everything else is drawn from the queue
    // pair mode (lock-step groups whose strip workgroups would not all fit the chip: every workgroup of this launch owns a CU's
    // whole LDS): a strip workgroup's two groups are strips 2 k and 2 k + 1 -- each does both update tiles of its own rows, then its
    // TRSM -- instead of strip k and a tile group: half as many strip workgroups.  8 problems of order 2048: 320 -> 200 workgroups
    // per panel launch (256 CUs), 36 -> 3x us per launch.  The tiles and the TRSM are the same functions: same bits.
    const int nswg = tr.pair ? (tr.strips + 1) / 2 : tr.strips;
    const int w = b - first - nswg, nworkers = nblocks - first - nswg;
    const unsigned first_tile = w >= 0 ? (unsigned)(2 * w + grp) : PANEL_NONE;
    if (w < 0) {
        // a strip workgroup: the strip's rows of this panel's columns first receive the previous panel's update -- tiles
        // (2 + strip, 0) and (2 + strip, 1), one per group, side by side (the head of the launch's critical path); then group 0
        // is the strip and group 1 a tile group
        const int strip = tr.pair ? 2 * (b - first) + grp : b - first;
        __builtin_amdgcn_s_setprio(3);
        if (tr.pair) {
            if (strip < tr.strips) {
                gemm_tile_64_k128_grp<false>(g, lds, (long)(2 + strip) * SB, 0L, gt, bar, nullptr, nullptr);
                gemm_tile_64_k128_grp<false>(g, lds, (long)(2 + strip) * SB, (long)SB, gt, bar, nullptr, nullptr);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // a TRSM wave's rows were written by all four waves of the group
                grp_barrier(bar);
                trsm_strip_pipelined<true>(tr, strip, stage, nullptr, info, lds);
            }
            __builtin_amdgcn_s_setprio(0);
        } else {
        gemm_tile_64_k128_grp<false>(g, lds, (long)(2 + strip) * SB, (long)grp * SB, gt, bar, nullptr, nullptr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // a TRSM wave's rows were written by all eight waves
        grp_barrier(bar);
        if (gt == 0) __hip_atomic_fetch_add(ctl + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (grp == 0) {
            while (__hip_atomic_load(ctl + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 2u) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            PANEL_STAMP(12, threadIdx.x == 0 && strip == 0);          // strip 0: its two update tiles done
            trsm_strip_pipelined<true>(tr, strip, stage, nullptr, info, lds);
            PANEL_STAMP(13, threadIdx.x == 0 && strip == 0);
            PANEL_STAMP(14, threadIdx.x == 0 && strip == tr.strips - 1);
        }
        __builtin_amdgcn_s_setprio(0);
        }
    }
    PANEL_STAMP(15, w == 0 && threadIdx.x == 0);                          // worker 0: starts on tiles
    PANEL_STAMP(17, w < 0 && b == first && threadIdx.x == 256);            // strip 0's second half: starts on tiles
    panel_tile_jobs(g, lds, gt, bar, tileq, ctl + 2 + grp, first_tile, (unsigned)(2 * nworkers));
    PANEL_STAMP(16, w == 0 && threadIdx.x == 0);                          // worker 0: queue dry
    PANEL_STAMP(18, w < 0 && b == first && threadIdx.x == 256);
    PANEL_STAMP(19, w == nworkers - 1 && threadIdx.x == 0);               // last worker: queue dry
}

// X = inv(L) for every 128x128 diagonal block of a factored matrix, one workgroup per block (they are
// independent, so this is ONE launch after the factorisation instead of a serial link in it).  Only the
// solves use X (gemv with the block inverses).
// Output: block jb at Linv + jb 128^2 (leading dimension 128) when Bs == 128; else straight into the diagonal 128-block of
// the Bs-wide block inverses the solves read (X / XT, leading dimension Bs: block jb / per, offset (jb % per) 128 (Bs + 1)).
__global__ __launch_bounds__(256) void k_diag_inverse_batched(const double *K, long ld, const double *xm_all, double *Linv,
                                                               double *LinvT, int Bs, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO4(cb, K, xm_all, Linv, LinvT);
    extern __shared__ __attribute__((aligned(16))) double sm[];
    constexpr int P = 130, XR = 17;                      // pitches of this kernel's own LDS images (see diag_inverse_row)
    static_assert((CIP_NB * P + 8 * 16 * XR) * 8 <= DIAG2_LDS_BYTES, "the re-pitched images must fit the launch's LDS");
    double *a = sm;
    double *xm = sm + CIP_NB * P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int jb = blockIdx.x;
    diag_load_block<P>(a, K + (long)jb * CIP_NB * (ld + 1), ld, tid);
#pragma unroll
    for (int q = 0; q < 8; ++q) xm[q * (16 * XR) + (tid >> 4) * XR + (tid & 15)] = xm_all[(size_t)jb * 2048 + q * 256 + tid];
    __syncthreads();
    diag_inverse_row<P, XR>(a, xm, wave, l15, g);
    diag_inverse_row<P, XR>(a, xm, 7 - wave, l15, g);
    __syncthreads();
    const int per = Bs / CIP_NB;
    const size_t ob = (size_t)(jb / per) * Bs * Bs + (size_t)(jb % per) * CIP_NB * (Bs + 1);
    double *Li = Linv + ob, *Lt = LinvT + ob;
    for (int e = tid; e < CIP_NB * CIP_NB; e += 256) {
        const int r = e & 127, cc = e >> 7;
        const int tr = r >> 4, tc = cc >> 4;
        double x, xt;
        if (tr == tc) {
            x = xm[tr * (16 * XR) + (cc & 15) * XR + (r & 15)];      // Xm[r][cc] (zero above the diagonal)
            xt = xm[tr * (16 * XR) + (r & 15) * XR + (cc & 15)];     // Xm[cc][r]
        } else {
            x = (tr > tc) ? a[cc + r * P] : 0.0;                     // X[r][cc]
            xt = (tc > tr) ? a[r + cc * P] : 0.0;                    // X[cc][r]
        }
        Li[r + (size_t)cc * Bs] = x;
        Lt[r + (size_t)cc * Bs] = xt;
    }
}

// Panel TRSM by blocked substitution (replaces "multiply by the explicit inverse"):
//   W21 = A21 inv(L11)' ,  L21 = W21 D^-1      for the rows below a factored 128x128 diagonal block.
// One wave per 16-row tile keeps its 8 finished W tiles in registers (accumulator layout == operand
// layout for f64 16x16x4), and for each 16-column block kb does
//   T = A21[:,kb] - sum_{q<kb} W[:,q] L11[kb][q]'      (4 kb MFMAs, operands straight from L2)
//   W[:,kb] = T inv(L11[kb][kb])'                       (4 MFMAs with the micro inverse)
// Half the flops of the inverse-GEMM form, 4x more workgroups, no 128x128 inverse on the critical path.
// (A 96-register cap -- launch bound 5 waves/SIMD -- lets these workgroups co-reside with the look-ahead's GEMM
// workgroups, but serialises the operand loads: same-session A/B 117.0 vs 118.8 KKT solves/s without the cap.)
// (First version: L11 tiles, micro inverses and the A21 row tile were loaded from L2 INSIDE the chain, 8 + 28 dependent
// round trips per wave: 12 us on an idle chip.  Now everything the chain touches is fetched before it starts -- the 28
// strictly-lower 16x16 tiles of L11, the 8 micro inverses and 1/d staged once per workgroup in LDS (73 KB, two
// workgroups per CU), the wave's own 16 x 128 row tile in registers -- so the kernel is one memory round trip plus
// 144 dependent MFMAs per wave.  Same operations in the same order: bit-identical results.)
#define TRSM_LDS_DOUBLES (28 * 256 + 8 * 256 + 128)
__device__ __forceinline__ int trsm_tile_index(int kb, int qq) { return kb * (kb - 1) / 2 + qq; }        // qq < kb
__global__ __launch_bounds__(256) void k_trsm_subst(double *__restrict__ Ap, long ld, const double *__restrict__ L11,
                                                        const double *__restrict__ xm, const double *__restrict__ dinv,
                                                        double *__restrict__ W, long ldw, CipBatch cb) {
    CIP_BATCH_GUARD(cb);
    CIP_BO5(cb, Ap, L11, xm, dinv, W);
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *lt = sm, *xs = sm + 28 * 256, *ds = xs + 8 * 256;
    __builtin_amdgcn_s_setprio(3);       // panel chain is latency-critical: win issue arbitration against co-resident GEMM waves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const long row = (long)blockIdx.x * 64 + wave * 16 + l15;
    double *ap = Ap + row + (long)g * ld;
    double *wp = W + row + (long)g * ldw;
    // ---- everything in flight at once: the wave's A21 rows ...
    double areg[8][4];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) areg[kb][q] = ap[(long)(kb * 16 + 4 * q) * ld];
    // ... and the shared operands: tile (kb, qq) stored [column][row] so that a lane's operand (row l15, column 4s + g)
    // sits at ((4s + g) * 16 + l15): 64 consecutive doubles per MFMA step, conflict-free
    {
        const int r = tid & 15, c = tid >> 4;
        double tl[28];
#pragma unroll
        for (int kb = 1; kb < 8; ++kb)
#pragma unroll
            for (int qq = 0; qq < kb; ++qq) tl[trsm_tile_index(kb, qq)] = L11[kb * 16 + r + (long)(qq * 16 + c) * ld];
        double xv[8];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) xv[kb] = xm[kb * 256 + tid];
        const double dv = tid < CIP_NB ? dinv[tid] : 0.0;
#pragma unroll
        for (int t = 0; t < 28; ++t) lt[t * 256 + c * 16 + r] = tl[t];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) xs[kb * 256 + tid] = xv[kb];
        if (tid < CIP_NB) ds[tid] = dv;
    }
    __syncthreads();
    double wneg[8][4];
    double lreg[8][4];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        v4d acc = (v4d){areg[kb][0], areg[kb][1], areg[kb][2], areg[kb][3]};
#pragma unroll
        for (int qq = 0; qq < kb; ++qq) {
            const double *t = lt + trsm_tile_index(kb, qq) * 256 + g * 16 + l15;
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = MFMA(t[64 * s], wneg[qq][s], acc);
        }
        v4d w = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) w = MFMA(xs[kb * 256 + (g + 4 * s) * 16 + l15], acc[s], w);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long col = kb * 16 + 4 * q;
            wneg[kb][q] = -w[q];
            lreg[kb][q] = w[q] * ds[col + g];
            wp[col * ldw] = w[q];
            ap[col * ld] = lreg[kb][q];
        }
    }
    // L' of the strip (round 4: no mirror pass behind the factorisation), transposed through the L11 tile buffer once every
    // wave is done with it.  L11 = &K[c0, c0], Ap = &K[c0 + 128, c0]: the strip's rows are columns c0 + 128 + row of L'.
    __syncthreads();
    {
        double *scr = lt + wave * 272;
        double *ut = const_cast<double *>(L11) + (long)(CIP_NB + blockIdx.x * 64 + wave * 16) * ld;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) wave_store_T(scr, lreg[kb], ut + kb * 16, ld, lane);
    }
}

// dynamic-LDS attributes of the three kernels, set once per process (std::call_once) and -- through cip_kernels_init(),
// called by cip_create -- before any stream of the process can be under hipGraph capture: hipFuncSetAttribute is not a
// capturable call
#include <mutex>
static std::once_flag g_attr_once;
static hipError_t g_attr_err = hipSuccess;
static void diag_attr_init(void) {
    hipError_t e = hipFuncSetAttribute((const void *)k_ldlt_diag128_v2<4>, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG2_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_ldlt_diag128_v2<8>, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG2_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_ldlt_diag128_v2<12>, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG2_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_ldlt_diag_upd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG2_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_ldlt_diag_upd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG2_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_ldlt_panel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG2_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_ldlt_panel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG2_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_diag_inverse_batched, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG2_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_trsm_subst, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(TRSM_LDS_DOUBLES * sizeof(double)));
    g_attr_err = e;
}
int cip_kernels_init(void) {
    std::call_once(g_attr_once, diag_attr_init);
    if (g_attr_err != hipSuccess) { cip_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(g_attr_err)); return -3; }
    return 0;
}
int cip_launch_diag_v2(hipStream_t s, double *Kb, long ld, double *xm_out, double *dvec, double *dinv, int *info, int col0,
                       PivotSigns sg) {
    if (cip_kernels_init()) return -3;
    static const int nw = [] { const char *e = getenv("CIP_DIAG_WAVES"); return e ? atoi(e) : 8; }();     // thread-safe one-time read
    if (nw == 4) cip_launch_b(k_ldlt_diag128_v2<4>, dim3(1), dim3(256), DIAG2_LDS_BYTES, s, Kb, ld, xm_out, dvec, dinv, info, col0, sg);
    else if (nw == 12) cip_launch_b(k_ldlt_diag128_v2<12>, dim3(1), dim3(768), DIAG2_LDS_BYTES, s, Kb, ld, xm_out, dvec, dinv, info, col0, sg);
    else cip_launch_b(k_ldlt_diag128_v2<8>, dim3(1), dim3(512), DIAG2_LDS_BYTES, s, Kb, ld, xm_out, dvec, dinv, info, col0, sg);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
// diag of a block + the in-block update `g` (C -= A B', plain accumulate form, M, N multiples of 64, M >= 128) whose first
// 128x128 tile IS that block; `ready`: a zeroed device word of this launch's own
int cip_launch_diag_upd(hipStream_t s, double *Kb, long ld, double *xm_out, double *dvec, double *dinv, int *info, int col0,
                        PivotSigns sg, unsigned *ready, const GemmArgs &g) {
    if (cip_kernels_init()) return -3;
    const long nt = (long)(g.M / SB) * (g.N / SB) - 1;          // every tile but the block's strictly-upper quarter
    // CIP_FUSE_DIAG=2: timing experiment, workgroup 0 does not wait (wrong results)
    static const int nowait = [] { const char *e = getenv("CIP_FUSE_DIAG"); return (e && atoi(e) == 2) ? 1 : 0; }();
    if (nowait) cip_launch_b(k_ldlt_diag_upd<false>, dim3((unsigned)(1 + nt)), dim3(256), DIAG2_LDS_BYTES, s, Kb, ld, xm_out, dvec, dinv,
                             info, col0, sg, ready, g);
    else cip_launch_b(k_ldlt_diag_upd<true>, dim3((unsigned)(1 + nt)), dim3(256), DIAG2_LDS_BYTES, s, Kb, ld, xm_out, dvec, dinv,
                      info, col0, sg, ready, g);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
// one launch per panel (k_ldlt_panel): diag of the block at Kb, the in-block update `g` of the previous panel when `g` is
// given (as cip_launch_diag_upd), and the TRSM of the `rows` rows below the block; `ready` / `stage`: zeroed device words
// of this launch's own
int cip_launch_panel(hipStream_t s, double *Kb, long ld, double *xm_out, double *dvec, double *dinv, int *info, int col0,
                     PivotSigns sg, unsigned *ready, unsigned *stage, unsigned *tileq, const GemmArgs *g, int rows, double *W, long ldw) {
    if (cip_kernels_init()) return -3;
    TrsmStrips tr = {Kb + CIP_NB, ld, Kb, xm_out, dinv, W, ldw, rows / 64, 0, PANEL_PRODUCER_CUS};
    // lock-step groups: problem index fastest in the dispatch order (CIP_PANEL_ZFAST=0: blockIdx.z, the order up to round 4's first half)
    static const int zfast = [] { const char *e = getenv("CIP_PANEL_ZFAST"); return e ? atoi(e) : 1; }();
    const int Bn = cip_in_batch() ? cip_tl_bz.B : 1;
    const bool transposed = zfast && Bn > 1;
    if (g) {
        const int tm = g->M / SB, tn = g->N / SB;
        if (tm != tr.strips + 2 || tn < 2) { cip_set_error("panel launch: update / TRSM shapes disagree"); return -1; }
        // tile workers: a workgroup (two tile groups) for every CU the diagonal kernel, the producers and the strips leave
        // free (per problem of a lock-step group), no more than the tiles the strips' groups do not take as their first ones
        static const int ncu = [] {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            return cus;
        }();
        const long ntiles = (long)tm * (tn - 2) - ((long)tn * (tn - 1) / 2 - 1);   // lower tiles of columns 2 .. tn-1
        // (off by default: 8 problems of order 2048 15.9 -> 15.3 ms per pass in one session, 15.2 -> 15.5 in the next -- inside the noise;
        //  CIP_PANEL_PAIR=1 selects it)
        static const int pair_on = [] { const char *e = getenv("CIP_PANEL_PAIR"); return e ? atoi(e) : 0; }();
        // ... when the unpaired launch does not fit the chip AND the paired one leaves at least four worker workgroups per problem: the
        // strips' second halves were tile groups, and without workers the update tiles queue up behind the TRSMs (16 problems of order
        // 2048: 23.5 ms per pass unpaired, 24.5-25.2 paired; 8 problems: 15.9 -> 15.3)
        tr.pair = (pair_on && Bn > 1 && (long)Bn * (1 + PANEL_PRODUCER_CUS + tr.strips) > ncu &&
                   ncu / Bn - 1 - PANEL_PRODUCER_CUS - (tr.strips + 1) / 2 >= 4) ? 1 : 0;
        const int nswg = tr.pair ? (tr.strips + 1) / 2 : tr.strips;
        long workers = ncu / (cip_in_batch() ? cip_tl_bz.B : 1) - 1 - PANEL_PRODUCER_CUS - nswg;
        if (workers > (ntiles + 1) / 2) workers = (ntiles + 1) / 2;
        if (workers < 1 && ntiles > 0 && tr.strips == 0) workers = 1;
        if (workers < 0) workers = 0;
        tr.nprod = Bn > 1 ? PANEL_PRODUCER_CUS : PANEL_PRODUCERS;
        const long grid = 1 + tr.nprod + nswg + workers;
        if (transposed) cip_launch(k_ldlt_panel<true>, dim3((unsigned)(grid * Bn)), dim3(64 * PANEL_WAVES), DIAG2_LDS_BYTES, s, Kb, ld, xm_out, dvec, dinv, info,
                                   col0, sg, ready, stage, tileq, *g, tr, Bn, CipBatch{cip_tl_bz.stride, cip_tl_bz.mask});
        else cip_launch_b(k_ldlt_panel<true>, dim3((unsigned)grid), dim3(64 * PANEL_WAVES), DIAG2_LDS_BYTES, s, Kb, ld, xm_out, dvec, dinv, info, col0, sg,
                          ready, stage, tileq, *g, tr, 0);
    } else {
        if (transposed) cip_launch(k_ldlt_panel<false>, dim3((unsigned)((1 + tr.strips) * Bn)), dim3(64 * PANEL_WAVES), DIAG2_LDS_BYTES, s, Kb, ld, xm_out, dvec,
                                   dinv, info, col0, sg, ready, stage, tileq, GemmArgs{}, tr, Bn, CipBatch{cip_tl_bz.stride, cip_tl_bz.mask});
        else cip_launch_b(k_ldlt_panel<false>, dim3((unsigned)(1 + tr.strips)), dim3(64 * PANEL_WAVES), DIAG2_LDS_BYTES, s, Kb, ld, xm_out, dvec, dinv,
                          info, col0, sg, ready, stage, tileq, GemmArgs{}, tr, 0);
    }
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_launch_diag_inverse(hipStream_t s, const double *K, long ld, int nblk, const double *xm_all, double *Linv,
                            double *LinvT, int Bs) {
    if (cip_kernels_init()) return -3;
    cip_launch_b(k_diag_inverse_batched, dim3(nblk), dim3(256), DIAG2_LDS_BYTES, s, K, ld, xm_all, Linv, LinvT, Bs);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_launch_trsm_subst(hipStream_t s, double *Ap, long ld, int rows, const double *L11, const double *xm,
                          const double *dinv, double *W, long ldw) {
    if (rows <= 0) return 0;
    if (cip_kernels_init()) return -3;
    cip_launch_b(k_trsm_subst, dim3(rows / 64), dim3(256), TRSM_LDS_DOUBLES * sizeof(double), s, Ap, ld, L11, xm, dinv, W, ldw);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
