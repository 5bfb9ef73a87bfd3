// fp64 MFMA "NT" GEMM for gfx950:  C (+)= alpha * A * B'   (all column-major).
//
// This one kernel carries every O(N^3) term of the KKT path:
//   * LDL' trailing update   C -= (L21 D) L21'      (lower tiles only, K = outer block)
//   * Schur formation        S  = Q + (A'F^-1)(A'F^-1)'   (EPI_SYRKQ)
// i.e. the work the reference does in src/kktsolvers.jl:32-35 (dense GEMMs + QR)
// and :289-295 (Schur + LU), re-designed for CDNA4.
//
// Design (MI355X):
//   * 128x128 C tile per 256-thread workgroup = 4 wave64s in a 2x2 grid, each wave
//     a 64x64 sub-tile = 4x4 v_mfma_f64_16x16x4_f64 accumulators (128 acc VGPRs).
//   * Both operands are "row-contiguous, k-strided" in memory (a panel column is
//     a contiguous run of rows), so a k-tile of 16 columns is staged as
//     lds[k][row]: global_load_dwordx4 (one wave reads 1 KB contiguous per k) ->
//     ds_write_b128, double-buffered, one barrier per k-tile.
//   * Fragments are read with ds_read_b128: lane c takes rows (2c, 2c+1) of a
//     32-row group, feeding two MFMA tiles per read; the row pitch is 1024 B
//     (== 0 mod 256 B), which is conflict-free for the b128 lane groups.
//   * The MFMA is issued "transposed" (A-operand = B rows, B-operand = A rows)
//     so that an accumulator's lane index runs along C's rows: the epilogue then
//     moves 16-byte double2 per lane, 256 B contiguous per 16 lanes.
//   * blockIdx is remapped so that each XCD (own L2) walks a contiguous run of
//     tiles (bijective variant of the xcd swizzle).
#include "cip_internal.h"
#include "cip_gemm_tile.h"
#include <stdlib.h>
#include <mutex>

#define LDS_TILE (CIP_KT * CIP_NB)        // doubles per operand per buffer (2048)

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// linear tile index -> (block row, block column): row-major over the lower triangle, or column-major rectangle
__device__ __forceinline__ void tile_coords(int t, int lower, int tm, int &bi, int &bj) {
    if (lower) {
        bi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((long)bi * (bi + 1) / 2 > t) --bi;
        while ((long)(bi + 1) * (bi + 2) / 2 <= t) ++bi;
        bj = t - (int)((long)bi * (bi + 1) / 2);
    } else {
        bi = t % tm;
        bj = t / tm;
    }
}

template <int EPI>
__device__ __forceinline__ void gemm_tile_128(GemmArgs &g, double *lds, int bi, int bj) {

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    const long i0 = (long)bi * CIP_NB, j0 = (long)bj * CIP_NB;
    const double *Ap = g.A + i0 + 2 * lane;
    const double *Bp = g.B + j0 + 2 * lane;

    v2d ra[4], rb[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long k = (long)kt * CIP_KT + q * 4 + wave;
            ra[q] = *(const v2d *)(Ap + k * g.lda);
            rb[q] = *(const v2d *)(Bp + k * g.ldb);
        }
    };
    auto lstore = [&](int buf) {
        double *la = lds + buf * (2 * LDS_TILE);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *(v2d *)(la + (q * 4 + wave) * CIP_NB + 2 * lane) = ra[q];
            *(v2d *)(la + LDS_TILE + (q * 4 + wave) * CIP_NB + 2 * lane) = rb[q];
        }
    };

    v4d acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

    const int KT = g.K / CIP_KT;
    gload(0);
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) gload(kt + 1);
        const double *la = lds + buf * (2 * LDS_TILE) + wm * 64 + 2 * l15;
        const double *lb = lds + buf * (2 * LDS_TILE) + LDS_TILE + wn * 64 + 2 * l15;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kk = ks * 4 + l4;
            const v2d fi0 = *(const v2d *)(la + kk * CIP_NB);
            const v2d fi1 = *(const v2d *)(la + kk * CIP_NB + 32);
            const v2d fj0 = *(const v2d *)(lb + kk * CIP_NB);
            const v2d fj1 = *(const v2d *)(lb + kk * CIP_NB + 32);
            const double fi[4] = {fi0.x, fi0.y, fi1.x, fi1.y};
            const double fj[4] = {fj0.x, fj0.y, fj1.x, fj1.y};
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj[tj], fi[ti], acc[ti][tj], 0, 0, 0);
        }
        if (kt + 1 < KT) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  lane holds, for tile (ti,tj), reg q:  C[row, col] with
    //   row = i0 + wm*64 + (ti>>1)*32 + 2*l15 + (ti&1)
    //   col = j0 + wn*64 + (tj>>1)*32 + 2*(l4 + 4q) + (tj&1)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long col = j0 + wn * 64 + (tj >> 1) * 32 + 2 * (l4 + 4 * q) + (tj & 1);
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const long row = i0 + wm * 64 + gi * 32 + 2 * l15;
                v2d val = (v2d){acc[2 * gi][tj][q], acc[2 * gi + 1][tj][q]};
                if (EPI == EPI_ACCUM) {
                    double *cp = g.C + row + col * g.ldc;
                    v2d c = g.overwrite ? (v2d){0.0, 0.0} : *(v2d *)cp;
                    c += g.alpha * val;
                    *(v2d *)cp = c;
                    if (g.Ct) { g.Ct[col + row * g.ldct] = c.x; g.Ct[col + (row + 1) * g.ldct] = c.y; }
                } else {   // EPI_SYRKQ
                    if (col < g.nvalid) {
                        if (row + 1 < g.nvalid) {
                            // Q keeps the caller's (possibly odd) leading dimension: scalar loads
                            const double *qp = g.Qin + row + col * g.ldq;
                            const v2d qv = (v2d){qp[0], qp[1]};
                            *(v2d *)(g.C + row + col * g.ldc) = qv + g.alpha * val;
                        } else if (row < g.nvalid) {
                            g.C[row + col * g.ldc] = g.Qin[row + col * g.ldq] + g.alpha * val.x;
                        }
                    }
                }
            }
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm_nt_128(GemmArgs g, CipBatch cb) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * LDS_TILE];   // 64 KB
    bool live;
    const unsigned oz = gemm_batch_prologue(g, cb, live);
    if (!live) return;
    if (EPI == EPI_ACCUM) gemm_own_batch(g, oz);       // batched problems: grid.y x grid.z
    int bi, bj;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), g.lower, g.M / CIP_NB, bi, bj);   // grid may cover only the first tiles
    gemm_tile_128<EPI>(g, lds, bi, bj);
}

__global__ __launch_bounds__(256, 4) void k_gemm_nt_64(GemmArgs g, CipBatch cb) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * CIP_KT * SB];   // 32 KB
    bool live;
    (void)gemm_batch_prologue(g, cb, live);
    if (!live) return;
    __builtin_amdgcn_s_setprio(3);       // skinny critical-path updates: priority over co-resident waves
    const int tm = g.M / SB;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    gemm_tile_64(g, lds, (long)(t % tm) * SB, (long)(t / tm) * SB);
}

// The LDL' trailing update C -= (L21 D) L21' on the lower triangle, in 64x64 quarter tiles (its own symbol so that
// profiles separate it from the skinny in-block updates): block b -> 128-tile b/4, quadrant b%4.
// 5 workgroups (20 waves) per CU; measured against the 128x128-tile kernel at 2 workgroups per CU:
// 55.0 vs 52.4 TFLOP/s at r = 8192, K = 512 and 53.1 vs 44.0 at K = 256 (tools/gemm_bench.hip, same session).
template <int EPI>
__global__ __launch_bounds__(256, 4) void k_ldlt_trailing_64(GemmArgs g, CipBatch cb) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * CIP_KT * SB];   // 32 KB
    bool live;
    (void)gemm_batch_prologue(g, cb, live);
    if (!live) return;
    __builtin_amdgcn_s_setprio(3);                     // measured: 58.0 vs 56.6 TFLOP/s without
    int bi, bj;
    tile_coords((int)(blockIdx.x >> 2), 1, g.M / CIP_NB, bi, bj);
    const int sub = blockIdx.x & 3;
    if (bi == bj && sub == 2) return;            // strictly-upper quadrant of a diagonal tile: never referenced
    gemm_tile_64<EPI, true>(g, lds, (long)bi * CIP_NB + (sub & 1) * SB, (long)bj * CIP_NB + (sub >> 1) * SB);
}

// Batched small products (block-inverse doubling): grid.y x grid.z independent problems, C = alpha A B' (overwrite)
__global__ __launch_bounds__(256, 4) void k_gemm_nt_64_batched(GemmArgs g, CipBatch cb) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * CIP_KT * SB];   // 32 KB
    bool live;
    const unsigned oz = gemm_batch_prologue(g, cb, live);
    if (!live) return;
    gemm_own_batch(g, oz);
    const int tm = g.M / SB;
    if (g.lower == 2 && (blockIdx.x % tm) > (blockIdx.x / tm)) return;          // "upper only": tiles strictly below the diagonal are not wanted
    if (g.lower == 3 && (blockIdx.x % tm) < (blockIdx.x / tm)) return;          // "lower only"
    gemm_tile_64<EPI_STORE>(g, lds, (long)(blockIdx.x % tm) * SB, (long)(blockIdx.x / tm) * SB);
}

// The same batched overwrite form with ONE 16x16 tile of C per workgroup, the k range split over its four waves, operands from
// global memory (L2) straight into the MFMA lanes -- lane l supplies row l % 16, k = l / 16 of its operand tile -- no LDS
// staging; the partial accumulators of waves 1..3 are added to wave 0's in a fixed order (the form of sdp_large.hip's
// k_gemm_nt_small).  For the block-inverse doubling (ldlt.hip): a level is a handful of h x h x h products, and a 64x64 tile walks
// its whole K = h on one CU -- 512 dependent MFMAs per wave at h = 512, 14 us of one CU's MFMA pipe behind a latency-bound
// k-loop, 19 us per launch on an idle chip -- while here the same product is (h / 16)^2 workgroups with 8 h / 128 MFMAs per
// wave.  K a multiple of 128, M and N of 16.  Register q of lane l holds C[i0 + l % 16, j0 + l / 16 + 4 q].
__global__ __launch_bounds__(256) void k_gemm_nt_16_batched(GemmArgs g, CipBatch cb) {
    __shared__ double red[3][4][64];
    bool live;
    const unsigned oz = gemm_batch_prologue(g, cb, live);
    if (!live) return;
    gemm_own_batch(g, oz);
    const int tm = g.M / 16;
    const long i0 = (long)(blockIdx.x % tm) * 16, j0 = (long)(blockIdx.x / tm) * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int kq = g.K >> 2;
    const double *a = g.A + i0 + l15 + (long)(wave * kq + l4) * g.lda;
    const double *b = g.B + j0 + l15 + (long)(wave * kq + l4) * g.ldb;
    v4d acc0 = (v4d){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
    for (int k = 0; k < kq; k += 32) {
        double av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = a[(long)(k + 4 * u) * g.lda];
            bv[u] = b[(long)(k + 4 * u) * g.ldb];
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[u], av[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[u + 1], av[u + 1], acc1, 0, 0, 0);
        }
    }
    acc0 += acc1;
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave - 1][q][lane] = acc0[q];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double c = g.alpha * (((acc0[q] + red[0][q][lane]) + red[1][q][lane]) + red[2][q][lane]);
            const long i = i0 + l15, j = j0 + l4 + 4 * q;
            g.C[i + j * g.ldc] = c;
            if (g.Ct) g.Ct[j + i * g.ldct] = c;
        }
    }
}

// Schur formation S = Q + Wt Wt' (lower tiles) in quarter tiles: the long-K (K = m) counterpart of the trailing update
template <bool GLDS>
__global__ __launch_bounds__(256, 4) void k_syrkq_64(GemmArgs g, CipBatch cb) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * CIP_KT * SB];   // 32 KB
    bool live;
    (void)gemm_batch_prologue(g, cb, live);
    if (!live) return;
    if (GLDS) __builtin_amdgcn_s_setprio(3);
    int bi, bj;
    tile_coords((int)(blockIdx.x >> 2), 1, g.M / CIP_NB, bi, bj);
    const int sub = blockIdx.x & 3;
    if (bi == bj && sub == 2) return;
    gemm_tile_64<EPI_SYRKQ, GLDS>(g, lds, (long)bi * CIP_NB + (sub & 1) * SB, (long)bj * CIP_NB + (sub >> 1) * SB);
}

// The same with few output tiles and a long K -- config 4: S = 1024 x 1024 from K = m = 32896, 136 quarter tiles on a chip with
// room for 1280 workgroups, 23 TFLOP/s.  Split-K: grid.y slices of the k range, each slice's product stored to its own image
// (EPI_STORE), then k_syrk_reduce adds the images in slice order to Qin: deterministic, no atomics.
__global__ __launch_bounds__(256, 4) void k_syrk_splitk_64(GemmArgs g, CipBatch cb) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * CIP_KT * SB];   // 32 KB
    bool live;
    (void)gemm_batch_prologue(g, cb, live);
    if (!live) return;
    int bi, bj;
    tile_coords((int)(blockIdx.x >> 2), 1, g.M / CIP_NB, bi, bj);
    const int sub = blockIdx.x & 3;
    if (bi == bj && sub == 2) return;
    const long k0 = (long)blockIdx.y * g.ksplit_len;
    g.A += k0 * g.lda; g.B += k0 * g.ldb;
    g.K = (g.K - k0 < g.ksplit_len) ? (int)(g.K - k0) : g.ksplit_len;
    g.C = (double *)((char *)g.ksplit_ws + (long)(blockIdx.z / (g.bz > 0 ? g.bz : 1)) * cb.stride) + (long)blockIdx.y * g.M * g.M;
    g.ldc = g.M; g.Ct = nullptr;
    gemm_tile_64<EPI_STORE>(g, lds, (long)bi * CIP_NB + (sub & 1) * SB, (long)bj * CIP_NB + (sub >> 1) * SB);
}
// The same on 128x128 tiles (two workgroups per CU): half the operand traffic per flop.  With K = m in the tens of thousands a
// slice's operand panel (M x len doubles) is far larger than an XCD's L2 and every tile streams it again: the 64-tile form ran
// at 36 TFLOP/s at M = 1024, K = 32896 (config 4).  CIP_SYRK_TILE = 64 / 128 forces a form.
__global__ __launch_bounds__(256, 2) void k_syrk_splitk_128(GemmArgs g, CipBatch cb) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * LDS_TILE];   // 64 KB
    bool live;
    (void)gemm_batch_prologue(g, cb, live);
    if (!live) return;
    int bi, bj;
    tile_coords((int)blockIdx.x, 1, g.M / CIP_NB, bi, bj);
    const long k0 = (long)blockIdx.y * g.ksplit_len;
    g.A += k0 * g.lda; g.B += k0 * g.ldb;
    g.K = (g.K - k0 < g.ksplit_len) ? (int)(g.K - k0) : g.ksplit_len;
    g.C = (double *)((char *)g.ksplit_ws + (long)(blockIdx.z / (g.bz > 0 ? g.bz : 1)) * cb.stride) + (long)blockIdx.y * g.M * g.M;
    g.ldc = g.M; g.Ct = nullptr; g.overwrite = 1;
    gemm_tile_128<EPI_ACCUM>(g, lds, bi, bj);
}
// C[i, j] = Qin[i, j] + sum_b image_b[i, j] for i >= j (by 64-tiles), i, j < nvalid; one thread per row pair of a 64 x 64 tile column
__global__ __launch_bounds__(256) void k_syrk_reduce(GemmArgs g, CipBatch cb) {
    bool live;
    (void)gemm_batch_prologue(g, cb, live);
    if (!live) return;
    const double *ws = (const double *)((const char *)g.ksplit_ws + (long)(blockIdx.z / (g.bz > 0 ? g.bz : 1)) * cb.stride);
    int bi, bj;
    tile_coords((int)blockIdx.x, 1, g.M / SB, bi, bj);                 // 64-tiles of the lower triangle
    const int r2 = threadIdx.x & 31, c0 = threadIdx.x >> 5;             // row pair, first column (8 columns per pass)
    const long row = (long)bi * SB + 2 * r2;
    const long img = (long)g.M * g.M;
    for (int c = c0; c < SB; c += 8) {
        const long col = (long)bj * SB + c;
        if (col >= g.nvalid || row >= g.nvalid) continue;
        v2d acc = (v2d){0.0, 0.0};
        for (int b = 0; b < g.ksplit_n; ++b) acc += *(const v2d *)(ws + b * img + row + col * g.M);
        const double *qp = g.Qin + row + col * g.ldq;
        double *cp = g.C + row + col * g.ldc;
        if (row + 1 < g.nvalid) *(v2d *)cp = (v2d){qp[0], qp[1]} + g.alpha * acc;
        else *cp = qp[0] + g.alpha * acc.x;
    }
}
// 128-tile form of the split: few tiles and a K so long that the 64-tile form is bound by re-streaming its operands
static bool syrk_split_128(int M, int K) {
    static const int force = [] { const char *e = getenv("CIP_SYRK_TILE"); return e ? atoi(e) : 0; }();
    if (force == 64) return false;
    const long tm = M / CIP_NB, t128 = tm * (tm + 1) / 2;
    if (force == 128) return t128 <= 256;
    return t128 <= 64 && K >= 16384;
}
int cip_syrk_split(int M, int K, int *len) {
    static const int on = [] { const char *e = getenv("CIP_SYRK_SPLITK"); return e ? atoi(e) : 1; }();
    const long tm = M / CIP_NB, wgs = 4 * (tm * (tm + 1) / 2);
    int n = 1;
    if (on && wgs < 640 && K >= 4096) {
        if (syrk_split_128(M, K)) { n = (int)(512 / (wgs / 4)); if (n > 16) n = 16; if (n < 1) n = 1; }     // 512 slots of 64 KB of LDS
        else { n = (int)((1280 + wgs - 1) / wgs); if (n > 16) n = 16; }
    }
    int l = ((K + n - 1) / n + CIP_KT - 1) / CIP_KT * CIP_KT;
    n = (K + l - 1) / l;
    if (len) *len = l;
    return n;
}

// XCD-aware order (optional, CIP_TRAIL_PATCH): workgroups are dealt round-robin to the 8 XCDs, so XCD x is handed whole
// p x p patches of quarter tiles (patches x, x+8, ...): the 2p half-panels of a patch are then fetched into that
// XCD's L2 once for p^2 tiles instead of once per tile.
__global__ __launch_bounds__(256, 4) void k_ldlt_trailing_64p(GemmArgs g, int psz, int npatch, int P, CipBatch cb) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * CIP_KT * SB];   // 32 KB
    bool live;
    (void)gemm_batch_prologue(g, cb, live);
    if (!live) return;
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x, x = b & 7, sq = b >> 3, per = psz * psz;
    const int patch = x + 8 * (sq / per), w = sq % per;
    if (patch >= npatch) return;
    int pi, pj;
    tile_coords(patch, 1, P, pi, pj);
    const int nq = g.M / SB;
    const int qi = pi * psz + w % psz, qj = pj * psz + w / psz;
    if (qi >= nq || qj >= nq) return;
    if ((qi >> 1) < (qj >> 1) || ((qi >> 1) == (qj >> 1) && qi < qj)) return;      // above the (128-tile) diagonal
    gemm_tile_64(g, lds, (long)qi * SB, (long)qj * SB);
}

int cip_launch_gemm(hipStream_t s, int epi, const GemmArgs &g) {
    if (g.M <= 0 || g.N <= 0) return 0;
    if (g.M % CIP_NB || g.N % CIP_NB || g.K % CIP_KT || g.K <= 0) {
        cip_set_error("gemm: bad dims M=%d N=%d K=%d", g.M, g.N, g.K);
        return -1;
    }
    const int tm = g.M / CIP_NB, tn = g.N / CIP_NB;
    long tiles;
    if (g.lower == 1) {
        if (g.M != g.N) { cip_set_error("gemm: lower needs M == N"); return -1; }
        tiles = (long)tm * (tm + 1) / 2;
    } else {
        tiles = (long)tm * tn;
    }
    const int by = g.by > 0 ? g.by : 1, bz = g.bz > 0 ? g.bz : 1;
    if (by * bz > 1 || (g.overwrite && epi == EPI_ACCUM && g.lower != 1)) {
        if (epi != EPI_ACCUM || g.lower == 1 || (g.lower >= 2 && !g.overwrite)) { cip_set_error("gemm: batching needs the plain accumulate form"); return -1; }
        if (g.overwrite && g.tiny16 && !g.lower && g.K % 128 == 0) {
            cip_launch_b(k_gemm_nt_16_batched, dim3((unsigned)((g.M / 16) * (g.N / 16)), by, bz), dim3(256), 0, s, g);
            CIP_HIP_CHECK(hipGetLastError());
            return 0;
        }
        if (g.overwrite) {
            cip_launch_b(k_gemm_nt_64_batched, dim3((unsigned)(4 * tiles), by, bz), dim3(256), 0, s, g);
            CIP_HIP_CHECK(hipGetLastError());
            return 0;
        }
        cip_launch_b(k_gemm_nt_128<EPI_ACCUM>, dim3((unsigned)tiles, by, bz), dim3(256), 0, s, g);
        CIP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    // CIP_GEMM_TILE=128: the lower-triangular update on the 128x128 kernel (kept for A/B runs of tools/gemm_bench.hip)
    static const int g_tile = [] { const char *e = getenv("CIP_GEMM_TILE"); return (e && atoi(e) == 128) ? 128 : 64; }();
    if (epi == EPI_LAZYC) {
        if (!g.lower || !g.Qin || !g.Cdiag || (g.ldq & 1) || (((uintptr_t)g.Qin) & 15)) { cip_set_error("gemm: bad lazy-C arguments"); return -1; }
        cip_launch_b(k_ldlt_trailing_64<EPI_LAZYC>, dim3((unsigned)(4 * tiles)), dim3(256), 0, s, g);
        CIP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (epi == EPI_ACCUM && g.lower && g_tile == 64) {
        // the LDL' trailing update: every 128-tile of the lower triangle as four 64x64 quarter tiles
        // Optional XCD-aware patch order (CIP_TRAIL_PATCH=4 or 8).  PMC at r = 8192, K = 512: 1.52 GB of L2-miss
        // fetches per launch in plain tile order, 0.94 GB with 4x4 patches, 0.75 GB with 8x8 -- but the launch is not
        // faster (standalone +1.5 % at r = 8192, -1 % at r = 2048) and the factorisation is 1-2 % SLOWER (same-session
        // A/B: 52.6 / 51.9 vs 51.0 / 50.4 TFLOP/s): the misses are served by the Infinity Cache and hidden by the 20
        // waves per CU, while a patch granularity costs load balance over the 8 XCDs.  Off by default.
        static const int psz = [] { const char *e = getenv("CIP_TRAIL_PATCH"); return e ? atoi(e) : 0; }();
        if (psz > 0) {
            const int nq = g.M / SB, P = (nq + psz - 1) / psz, npatch = P * (P + 1) / 2;
            const long grid = (long)((npatch + 7) / 8) * 8 * psz * psz;
            cip_launch_b(k_ldlt_trailing_64p, dim3((unsigned)grid), dim3(256), 0, s, g, psz, npatch, P);
            CIP_HIP_CHECK(hipGetLastError());
            return 0;
        }
        cip_launch_b(k_ldlt_trailing_64<EPI_ACCUM>, dim3((unsigned)(4 * tiles)), dim3(256), 0, s, g);
        CIP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (epi == EPI_SYRKQ && g.lower && g_tile == 64 && g.ksplit_ws && g.ksplit_n > 1) {
        GemmArgs gs = g;
        gs.alpha = 1.0;                                          // (the images hold the plain products; alpha is applied by the reduction)
        if (syrk_split_128(g.M, g.K)) cip_launch_b(k_syrk_splitk_128, dim3((unsigned)tiles, (unsigned)g.ksplit_n), dim3(256), 0, s, gs);
        else cip_launch_b(k_syrk_splitk_64, dim3((unsigned)(4 * tiles), (unsigned)g.ksplit_n), dim3(256), 0, s, gs);
        const long t64 = (long)(g.M / SB) * (g.M / SB + 1) / 2;
        cip_launch_b(k_syrk_reduce, dim3((unsigned)t64), dim3(256), 0, s, g);
        CIP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (epi == EPI_SYRKQ && g.lower && g_tile == 64) {
        // operands global -> LDS directly + raised wave priority, as the trailing update (round 4, config 3: 1.35 -> 1.28 ms per
        // Schur formation, same-session A/B 4.17 -> 4.10 ms per iteration, same bits); CIP_SYRK_GLDS=0: register staging
        static const int glds = [] { const char *e = getenv("CIP_SYRK_GLDS"); return e ? atoi(e) : 1; }();
        if (glds && !(g.lda & 1) && !(g.ldb & 1) && !(((uintptr_t)g.A | (uintptr_t)g.B) & 15)) cip_launch_b(k_syrkq_64<true>, dim3((unsigned)(4 * tiles)), dim3(256), 0, s, g);
        else cip_launch_b(k_syrkq_64<false>, dim3((unsigned)(4 * tiles)), dim3(256), 0, s, g);
        CIP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (epi == EPI_ACCUM && !g.lower && !g.overwrite && (tiles < 256 || g.force64) && g.M % SB == 0 && g.N % SB == 0) {
        // skinny, latency-critical: quarter-size tiles
        const long t64 = (long)(g.M / SB) * (g.N / SB);
        cip_launch_b(k_gemm_nt_64, dim3((unsigned)t64), dim3(256), 0, s, g);
        CIP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    dim3 grid((unsigned)tiles), block(256);
    switch (epi) {
        case EPI_ACCUM: cip_launch_b(k_gemm_nt_128<EPI_ACCUM>, grid, block, 0, s, g); break;
        case EPI_SYRKQ: cip_launch_b(k_gemm_nt_128<EPI_SYRKQ>, grid, block, 0, s, g); break;
        default: cip_set_error("gemm: bad epilogue"); return -1;
    }
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
