// Device pieces of the fp64 MFMA GEMM shared by gemm_f64.hip and the fused diagonal + in-block-update kernel of diag.hip:
// the 64x64 tile computation (gemm_tile_64) and the lock-step batch prologue of GEMM kernels.
#pragma once
#include "cip_internal.h"

// Lock-step batches (cip_internal.h): grid.z = (own batch count, >= 1) x (problems); returns the launch's own z index
// after shifting every operand pointer by the problem's slab offset; live = false: the problem is masked off.
__device__ __forceinline__ unsigned gemm_batch_prologue(GemmArgs &g, const CipBatch &cb, bool &live) {
    const unsigned gz = g.bz > 0 ? (unsigned)g.bz : 1u;
    const unsigned pz = blockIdx.z / gz;
    live = ((cb.mask >> pz) & 1ull) != 0;
    const long off = (long)pz * cb.stride;
    g.A = (const double *)((const char *)g.A + off);
    g.B = (const double *)((const char *)g.B + off);
    g.C = (double *)((char *)g.C + off);
    if (g.Ct) g.Ct = (double *)((char *)g.Ct + off);
    if (g.Qin) g.Qin = (const double *)((const char *)g.Qin + off);
    if (g.Cdiag) g.Cdiag = (const double *)((const char *)g.Cdiag + off);
    return blockIdx.z - pz * gz;
}
// the launch's own (grid.y, grid.z) batching: pointer strides in doubles
__device__ __forceinline__ void gemm_own_batch(GemmArgs &g, unsigned oz) {
    g.A += blockIdx.y * g.sAy + oz * g.sAz;
    g.B += blockIdx.y * g.sBy + oz * g.sBz;
    g.C += blockIdx.y * g.sCy + oz * g.sCz;
    if (g.Ct) g.Ct += blockIdx.y * g.sCty + oz * g.sCtz;
}

// ---------------------------------------------------------------------------------------------
// Small-tile variant (64x64 C tile, wave = 32x32 = 2x2 MFMA tiles) for the latency-critical skinny
// updates on the factorisation's critical path (in-block strip update):
// 4x the workgroups and a quarter of the per-tile latency of the 128x128 kernel, at twice the LDS
// traffic per flop -- these launches carry < 10 % of the flops.  Accumulate epilogue only.
#define SB 64
// GLDS: operands go global -> LDS directly (`global_load_lds_dwordx4`: no staging registers, no ds_write pass).  The
// [k][64 rows] LDS image is lane-linear for the staging pattern below -- a wave's 64 x 16 bytes are two consecutive k columns --
// so the same image is produced.  Used by the trailing update: same-session A/B at n = 8192 54.0 -> 56.6 TFLOP/s (the
// kernel drops from 102 to 81 VGPRs and the ds_write pass of every k-tile); the K = 128 in-block tiles and the batched /
// SYRK forms measured no different with it and keep register staging.
template <int EPI = EPI_ACCUM, bool GLDS = false>
__device__ __forceinline__ void gemm_tile_64(const GemmArgs &g, double *lds, long i0, long j0) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    // staging: 64 rows x 16 k per operand = 512 double2 -> 2 per thread: e = q*256 + tid, k = e >> 5, rp = e & 31
    const int k_ld = tid >> 5, rp = tid & 31;
    const double *Ap = g.A + i0 + 2 * rp;
    const double *Bp = g.B + j0 + 2 * rp;
    v2d ra[2], rb[2];
    auto gload = [&](int kt) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const long k = (long)kt * CIP_KT + q * 8 + k_ld;
            ra[q] = *(const v2d *)(Ap + k * g.lda);
            rb[q] = *(const v2d *)(Bp + k * g.ldb);
        }
    };
    auto lstore = [&](int buf) {
        double *la = lds + buf * (2 * CIP_KT * SB);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *(v2d *)(la + (q * 8 + k_ld) * SB + 2 * rp) = ra[q];
            *(v2d *)(la + CIP_KT * SB + (q * 8 + k_ld) * SB + 2 * rp) = rb[q];
        }
    };
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int KT = g.K / CIP_KT;
    auto gdma = [&](int kt, int buf) {
        typedef __attribute__((address_space(1))) const void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
        double *la = lds + buf * (2 * CIP_KT * SB) + 2 * wave * SB;      // wave-uniform; the hardware adds lane x 16 bytes
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const long k = (long)kt * CIP_KT + q * 8 + k_ld;
            __builtin_amdgcn_global_load_lds((gptr_t)(Ap + k * g.lda), (lptr_t)(la + q * 8 * SB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(Bp + k * g.ldb), (lptr_t)(la + CIP_KT * SB + q * 8 * SB), 16, 0, 0);
        }
    };
    if (GLDS) gdma(0, 0);
    else { gload(0); lstore(0); }
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) { if (GLDS) gdma(kt + 1, buf ^ 1); else gload(kt + 1); }
        const double *la = lds + buf * (2 * CIP_KT * SB) + wm * 32 + 2 * l15;
        const double *lb = lds + buf * (2 * CIP_KT * SB) + CIP_KT * SB + wn * 32 + 2 * l15;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kk = ks * 4 + l4;
            const v2d fi = *(const v2d *)(la + kk * SB);
            const v2d fj = *(const v2d *)(lb + kk * SB);
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.x, fi.x, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.x, fi.y, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.y, fi.x, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.y, fi.y, acc[1][1], 0, 0, 0);
        }
        if (!GLDS && kt + 1 < KT) lstore(buf ^ 1);
        __syncthreads();
    }
    // lane holds, for tile (ti,tj), reg q: row = i0 + wm*32 + 2*l15 + ti, col = j0 + wn*32 + 2*(l4 + 4q) + tj
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long col = j0 + wn * 32 + 2 * (l4 + 4 * q) + tj;
            const long row = i0 + wm * 32 + 2 * l15;
            double *cp = g.C + row + col * g.ldc;
            const v2d val = (v2d){acc[0][tj][q], acc[1][tj][q]};
            if (EPI == EPI_ACCUM) {
                v2d c = *(v2d *)cp;
                c += g.alpha * val;
                *(v2d *)cp = c;
            } else if (EPI == EPI_LAZYC) {    // the C operand comes from Qin (+ Cdiag on the diagonal): what a copy into C would have put there
                v2d c = *(const v2d *)(g.Qin + row + col * g.ldq);
                if (row == col) c.x = g.Cdiag[row];
                if (row + 1 == col) c.y = g.Cdiag[col];
                c += g.alpha * val;
                *(v2d *)cp = c;
            } else if (EPI == EPI_STORE) {    // C = alpha acc, optionally also stored transposed
                const v2d c = g.alpha * val;
                *(v2d *)cp = c;
                if (g.Ct) { g.Ct[col + row * g.ldct] = c.x; g.Ct[col + (row + 1) * g.ldct] = c.y; }
            } else if (col < g.nvalid) {      // EPI_SYRKQ: C = Qin + alpha acc inside the valid n x n corner
                if (row + 1 < g.nvalid) {
                    const double *qp = g.Qin + row + col * g.ldq;       // Q keeps the caller's (possibly odd) pitch
                    *(v2d *)cp = (v2d){qp[0], qp[1]} + g.alpha * val;
                } else if (row < g.nvalid) {
                    *cp = g.Qin[row + col * g.ldq] + g.alpha * val.x;
                }
            }
        }
}


// K = 128 with the WHOLE of both operands resident in LDS (128 KB): one global round trip, one barrier, then 32 MFMA
// steps -- for launches that own a CU's LDS anyway (diag.hip: k_ldlt_diag_upd).  The k-loop of gemm_tile_64 waits for a
// global load in each of its eight iterations (~1 us each when the tile is alone on its CU).  Same accumulation order
// as gemm_tile_64: bit-identical results.  Accumulate epilogue.  SC1C: the C tile is written through (`sc1`): its reader
// is another workgroup of the same launch (k_ldlt_diag_upd).
template <bool SC1C>
__device__ __forceinline__ void gemm_tile_64_k128(const GemmArgs &g, double *lds, long i0, long j0) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int k_ld = tid >> 5, rp = tid & 31;
    const double *Ap = g.A + i0 + 2 * rp;
    const double *Bp = g.B + j0 + 2 * rp;
    double *la = lds, *lb = lds + 128 * SB;                     // [k][row], 128 x 64 each
    // the C tile is fetched FIRST, beside the operands: nobody else writes it in this launch and earlier launches are
    // complete, so plain 16-byte loads do; only the stores of an SC1C tile must be write-through (its reader is another CU)
    const long rowo = i0 + wm * 32 + 2 * l15;
    v2d cpre[2][4];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int q = 0; q < 4; ++q) cpre[tj][q] = *(const v2d *)(g.C + rowo + (j0 + wn * 32 + 2 * (l4 + 4 * q) + tj) * g.ldc);
    {
        v2d ra[16], rb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const long k = q * 8 + k_ld;
            ra[q] = *(const v2d *)(Ap + k * g.lda);
            rb[q] = *(const v2d *)(Bp + k * g.ldb);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            *(v2d *)(la + (q * 8 + k_ld) * SB + 2 * rp) = ra[q];
            *(v2d *)(lb + (q * 8 + k_ld) * SB + 2 * rp) = rb[q];
        }
    }
    __syncthreads();
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    const double *pa = la + wm * 32 + 2 * l15, *pb = lb + wn * 32 + 2 * l15;
#pragma unroll 8
    for (int ks = 0; ks < 32; ++ks) {
        const int kk = ks * 4 + l4;
        const v2d fi = *(const v2d *)(pa + kk * SB);
        const v2d fj = *(const v2d *)(pb + kk * SB);
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.x, fi.x, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.x, fi.y, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.y, fi.x, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.y, fi.y, acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double *cp = g.C + rowo + (j0 + wn * 32 + 2 * (l4 + 4 * q) + tj) * g.ldc;
            v2d c = cpre[tj][q];
            c += g.alpha * (v2d){acc[0][tj][q], acc[1][tj][q]};
            if (SC1C) {
                // one 16-byte write-through store (buffer store with the sc1 policy bit = agent scope)
                typedef int v4i_t __attribute__((ext_vector_type(4)));
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)g.C, 0, 0x7fffffff, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_t, c), rs, (int)((cp - g.C) * 8), 0, 16);
            } else {
                *(v2d *)cp = c;
            }
        }
}

// ---- the same K = 128 tile for a GROUP of four waves that shares its workgroup with another group (diag.hip: k_ldlt_panel,
// where a TRSM strip or a second tile job lives in the other four waves of the workgroup): no s_barrier -- the hardware
// barrier counts every live wave of the workgroup -- but a generation-counted LDS word the group's four waves meet on.
// LDS traffic of one wave is in order, so "my ds_writes have landed (lgkmcnt(0)), then my ds_add" is a release and the
// spin's ds_read followed by dependent ds_reads an acquire; the asm memory clobbers keep the compiler from moving LDS
// accesses across.  64 KB of LDS per group: the operands pass through two 32-KB buffers a QUARTER of K at a time (quarter
// q + 1 is written while quarter q feeds the MFMAs; the wait that publishes it also says everybody is done with the buffer
// quarter q + 2 goes into): four meetings per tile.  Same k order from a zero accumulator, same C + alpha acc: bit-identical
// to gemm_tile_64_k128.
struct GrpBar { unsigned *ctr; unsigned gen; };
__device__ __forceinline__ void grp_barrier(GrpBar &b) {
    b.gen += 4u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(b.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(b.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < b.gen) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
// QUEUE: the group works through a tile queue (an agent-scope counter): the index of its NEXT tile is drawn behind this
// tile's operand loads (loads return in order: the draw must not sit in front of them) and handed to the other three waves
// through `slot` at the tile's last meeting; returned to every thread.
template <bool QUEUE>
__device__ __forceinline__ unsigned gemm_tile_64_k128_grp(const GemmArgs &g, double *lds, long i0, long j0, int gt, GrpBar &bar,
                                                          unsigned *tileq, unsigned *slot) {
    const int lane = gt & 63;
    const int wave = gt >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int k_ld = gt >> 5, rp = gt & 31;
    const double *Ap = g.A + i0 + 2 * rp;
    const double *Bp = g.B + j0 + 2 * rp;
    const long rowo = i0 + wm * 32 + 2 * l15;
    v2d cpre[2][4];
    v2d ra[16], rb[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const long k = q * 8 + k_ld;
        ra[q] = *(const v2d *)(Ap + k * g.lda);
        rb[q] = *(const v2d *)(Bp + k * g.ldb);
    }
    unsigned nxt = 0u;
    if (QUEUE && gt == 0) nxt = __hip_atomic_fetch_add(tileq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        double *la = lds + (qt & 1) * (64 * SB), *lb = la + 32 * SB;       // buffer qt & 1: [32 k][64 rows] per operand
        if (qt == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *(v2d *)(la + (q * 8 + k_ld) * SB + 2 * rp) = ra[q];
                *(v2d *)(lb + (q * 8 + k_ld) * SB + 2 * rp) = rb[q];
            }
        }
        if (QUEUE && qt == 3 && gt == 0) *(volatile unsigned *)slot = nxt;
        grp_barrier(bar);
        if (qt == 1) {
            // the C tile: fetched once half of the staging registers are free again, two quarters of MFMAs ahead of its use
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int q = 0; q < 4; ++q) cpre[tj][q] = *(const v2d *)(g.C + rowo + (j0 + wn * 32 + 2 * (l4 + 4 * q) + tj) * g.ldc);
        }
        if (qt < 3) {
            double *na = lds + ((qt + 1) & 1) * (64 * SB), *nb = na + 32 * SB;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *(v2d *)(na + (q * 8 + k_ld) * SB + 2 * rp) = ra[4 * (qt + 1) + q];
                *(v2d *)(nb + (q * 8 + k_ld) * SB + 2 * rp) = rb[4 * (qt + 1) + q];
            }
        }
        const double *pa = la + wm * 32 + 2 * l15, *pb = lb + wn * 32 + 2 * l15;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int kk = ks * 4 + l4;
            const v2d fi = *(const v2d *)(pa + kk * SB);
            const v2d fj = *(const v2d *)(pb + kk * SB);
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.x, fi.x, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.x, fi.y, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.y, fi.x, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.y, fi.y, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double *cp = g.C + rowo + (j0 + wn * 32 + 2 * (l4 + 4 * q) + tj) * g.ldc;
            v2d c = cpre[tj][q];
            c += g.alpha * (v2d){acc[0][tj][q], acc[1][tj][q]};
            *(v2d *)cp = c;
        }
    if (QUEUE) nxt = *(volatile unsigned *)slot;
    return nxt;
}
