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
template <bool SCALEA>
__global__ __launch_bounds__(256, 4) void k_ldlt_trailing_64(GemmArgs g, CipBatch cb) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * CIP_KT * SB];   // 32 KB
    bool live;
    (void)gemm_batch_prologue(g, cb, live);
    if (!live) return;
    if (g.lowprio) __builtin_amdgcn_s_setprio(0);      // the bulk of the two-stream schedule: the chain's waves win issue arbitration
    else __builtin_amdgcn_s_setprio(3);                // measured: 58.0 vs 56.6 TFLOP/s without
    int bi, bj;
    tile_coords((int)(blockIdx.x >> 2), 1, g.M / CIP_NB, bi, bj);
    const int sub = blockIdx.x & 3;
    if (bi == bj && sub == 2) return;            // strictly-upper quadrant of a diagonal tile: never referenced
    // SCALEA: A = L (from K) scaled by d on its way into LDS -- the operand form of the look-ahead schedule's workers,
    // on the serial schedule (bit-identical results: the cross-check of the in-launch hand-offs, tests/test_gpu_lookahead.py)
    gemm_tile_64<EPI_ACCUM, false, SCALEA, !SCALEA>(g, lds, (long)bi * CIP_NB + (sub & 1) * SB, (long)bj * CIP_NB + (sub >> 1) * SB, g.dk);
}

// Batched small products (block-inverse doubling): grid.y x grid.z independent problems, C = alpha A B' (overwrite)
__global__ __launch_bounds__(256, 4) void k_gemm_nt_64_batched(GemmArgs g, CipBatch cb) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * CIP_KT * SB];   // 32 KB
    bool live;
    const unsigned oz = gemm_batch_prologue(g, cb, live);
    if (!live) return;
    gemm_own_batch(g, oz);
    const int tm = g.M / SB;
    gemm_tile_64<EPI_STORE>(g, lds, (long)(blockIdx.x % tm) * SB, (long)(blockIdx.x / tm) * SB);
}

// Schur formation S = Q + Wt Wt' (lower tiles) in quarter tiles: the long-K (K = m) counterpart of the trailing update
__global__ __launch_bounds__(256, 4) void k_syrkq_64(GemmArgs g, CipBatch cb) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * CIP_KT * SB];   // 32 KB
    bool live;
    (void)gemm_batch_prologue(g, cb, live);
    if (!live) return;
    int bi, bj;
    tile_coords((int)(blockIdx.x >> 2), 1, g.M / CIP_NB, bi, bj);
    const int sub = blockIdx.x & 3;
    if (bi == bj && sub == 2) return;
    gemm_tile_64<EPI_SYRKQ>(g, lds, (long)bi * CIP_NB + (sub & 1) * SB, (long)bj * CIP_NB + (sub >> 1) * SB);
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

// CU reservation for the look-ahead schedule.  The panel chain of the next outer blocks runs beside the persistent
// trailing-update launch; its diagonal kernel needs a CU's entire LDS and would otherwise wait until a CU has drained.
// Workgroups are dealt round-robin to the XCDs and, inside an XCD, to the shader engines whatever their occupancy
// (measured: with the reservation in one SE only, the single-workgroup diagonal kernel started at once in one launch
// out of four), so EVERY (XCD, SE) pair keeps `reserve` (1 or 2) CUs free: persistent workgroups that find themselves
// on a reserved CU exit at once.  CU ids differ per SE (harvesting); the ids are found once by a probe launch.
// Placement is used for speed only: any workgroup can take any tile.
struct ReserveMap { unsigned char cu[32][2]; };       // [xcc*4 + se][k] = k-th lowest CU id present (0xff: none)
__global__ void k_probe_cus(unsigned *out) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID: cu [11:8], se [15:13]
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0x7;  // HW_REG_XCC_ID
        out[blockIdx.x] = (xcc << 8) | (((hw >> 13) & 0x3) << 4) | ((hw >> 8) & 0xf);
    }
    const long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 100000) __builtin_amdgcn_s_sleep(10);   // keep the CU busy: ~50 us of shader clock
}
static ReserveMap g_rmap;
static int g_rmap_ready = 0;
static std::mutex g_rmap_mutex;
static int reserve_map_init(void) {
    std::lock_guard<std::mutex> lock(g_rmap_mutex);
    if (g_rmap_ready) return 0;
    const int nb = 1024;
    unsigned *d = nullptr;
    CIP_HIP_CHECK(hipMalloc(&d, nb * sizeof(unsigned)));
    hipLaunchKernelGGL(k_probe_cus, dim3(nb), dim3(256), 65536, 0, d);           // 64 KB of LDS each: 2 per CU
    CIP_HIP_CHECK(hipGetLastError());
    unsigned h[1024];
    CIP_HIP_CHECK(hipMemcpy(h, d, nb * sizeof(unsigned), hipMemcpyDeviceToHost));
    CIP_HIP_CHECK(hipFree(d));
    unsigned present[32] = {0};
    for (int b = 0; b < nb; ++b) present[((h[b] >> 8) & 7) * 4 + ((h[b] >> 4) & 3)] |= 1u << (h[b] & 0xf);
    for (int gse = 0; gse < 32; ++gse) {
        int k = 0;
        g_rmap.cu[gse][0] = g_rmap.cu[gse][1] = 0xff;
        for (int c = 0; c < 16 && k < 2; ++c)
            if (present[gse] & (1u << c)) g_rmap.cu[gse][k++] = (unsigned char)c;
    }
    g_rmap_ready = 1;
    return 0;
}
// ---------------------------------------------------------------------------------------------
// Deep look-ahead: ONE persistent launch carries every trailing update of a factorisation (ldlt.hip: factor_lookahead).
//
// Round J = the update with outer block J:  C[i, j] -= (L[i, J] D_J) L[j, J]'  for all 64x64 tiles (i >= j) of the
// trailing matrix.  The panel chain of block J+1 (diag / TRSM / in-block kernels, a serial chain of small launches on a
// high-priority stream, on CUs this kernel leaves free) needs only the column strip of block J+1 updated, so each round
// is split into a CRITICAL strip (the next outer block's columns) and the BULK (everything to its right), in two
// queues: a free worker always serves the critical queue first.  The chain therefore never waits for the bulk of any
// round (the first-generation look-ahead put strip and bulk on one in-order stream: one round of overlap at most),
// and the bulk of round J runs beside the chain of blocks J+1, J+2, ...
//
// Dependencies, all through device memory inside the launch:
//   flag[J]        set by a tiny kernel at the end of chain J on the chain's stream: L[:, J] and d_J are final.  A worker
//                  entering round J polls it (`sc1`), then ONE wave does an agent-scope acquire (invalidates the CU's L1)
//                  before the panel loads (MI355X_MICROARCH.md: consumer = poll -> acquire -> barrier -> plain loads).
//   done[tile]     rounds completed on that 64x64 tile of K: round J of a tile follows round J-1 of the same tile,
//                  possibly on another CU / XCD -> the C tile is read with `sc1` loads and written through with `sc1`
//                  stores, every storing wave drains (`vmcnt(0)`), barrier, one lane publishes done = J+1.
//   stripdone[S]   completed tile-rounds in the columns of outer block S; the chain of block S starts behind a gate
//                  kernel that waits for S * tiles(S).
// A critical tile of round J is handed out only when every round-(J-1) tile of the same strip has been HANDED OUT
// (bulk queue head beyond that strip), and bulk tiles are handed out in round order: every wait is for a tile some
// running workgroup already owns, whatever the number of workers.
#define LA_MAX_ROUNDS 64
// Queue geometry, recomputed on the fly by the scheduling lane (at most nrounds iterations of integer arithmetic per
// tile: no table in LDS -- the GEMM's 32 KB are all a workgroup may use if five are to share a CU).
struct LaGeom { int nrounds, nbo, nt, swq; };          // nt = Npad / 64, swq = nbo / 64
__host__ __device__ inline int la_crit_count(const LaGeom &g, int J) {            // tiles of round J's critical strip
    const int nq = g.nt - (J + 1) * g.swq, sw = nq < g.swq ? nq : g.swq;
    return sw * nq - sw * (sw - 1) / 2;
}
__host__ __device__ inline int la_bulk_count(const LaGeom &g, int J) {            // tiles right of it
    const int nq = g.nt - (J + 1) * g.swq, sw = nq < g.swq ? nq : g.swq, n2 = nq - sw;
    return n2 * (n2 + 1) / 2;
}
__host__ __device__ inline int la_bulk_first_strip(const LaGeom &g, int J) {      // ... of which in round J+1's critical strip
    const int nq = g.nt - (J + 1) * g.swq, sw = nq < g.swq ? nq : g.swq, n2 = nq - sw, s2 = n2 < g.swq ? n2 : g.swq;
    return s2 * n2 - s2 * (s2 - 1) / 2;
}
struct LaCtrl {                                 // device, zeroed before every factorisation
    unsigned crit_next, bulk_next;
    int err;
    unsigned workers_done;
    unsigned long long busy_ticks, tiles;       // s_memtime ticks (shader cycles) inside tile computations, summed over workers
    unsigned pad[8];
    unsigned flag[LA_MAX_ROUNDS + 1];
    unsigned stripdone[LA_MAX_ROUNDS + 1];
    // followed by done[nt * nt]
};

__device__ __forceinline__ unsigned la_load(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// bounded spin (thread 0 only): returns false after ~1 s and raises ctrl->err, so a logic error cannot hang the GPU
template <int SLEEP>
__device__ __forceinline__ bool la_wait_ge(const unsigned *p, unsigned target, LaCtrl *ctrl) {
    const long t0 = __builtin_amdgcn_s_memtime();
    while (la_load(p) < target) {
        __builtin_amdgcn_s_sleep(SLEEP);
        if (__builtin_amdgcn_s_memtime() - t0 > 2000000000L || la_load((const unsigned *)&ctrl->err) != 0) { atomicExch(&ctrl->err, -7); return false; }
    }
    return true;
}

// Scheduler (one lane per workgroup): next tile for this worker as (round, ci, cj) in global 64-tile coordinates;
// round = -2: nothing left (or the launch is being abandoned).  Critical queue first; see the protocol above.
__device__ __forceinline__ int3 la_next_tile(const LaGeom &g, int total_crit, int total_bulk, LaCtrl *ctrl, const unsigned *done) {
    int round = -1, ci = 0, cj = 0;
    for (int spins = 0;; ++spins) {
        // ---- critical queue first
        const int c = (int)la_load(&ctrl->crit_next);
        if (c < total_crit) {
            int Jc = 0, start = 0;
            for (int n; c >= start + (n = la_crit_count(g, Jc)); ++Jc) start += n;
            bool ready = la_load(&ctrl->flag[Jc]) != 0;
            if (ready && Jc > 0) {
                // every round-(Jc-1) tile of this strip must have been handed out (it sits at the head of that round's bulk list)
                int bs = 0;
                for (int J = 0; J < Jc - 1; ++J) bs += la_bulk_count(g, J);
                ready = (int)la_load(&ctrl->bulk_next) >= bs + la_bulk_first_strip(g, Jc - 1);
            }
            if (ready) {
                const int t = (int)atomicAdd(&ctrl->crit_next, 1u);
                if (t < total_crit) {
                    int J = Jc;
                    for (int n; t >= start + (n = la_crit_count(g, J)); ++J) start += n;
                    const int r0q = (J + 1) * g.swq, nq = g.nt - r0q;                 // trailing matrix in 64-tiles
                    int idx = t - start;
                    cj = 0;
                    while (idx >= nq - cj) { idx -= nq - cj; ++cj; }                  // column-major, rows cj .. nq-1
                    ci = cj + idx + r0q; cj += r0q; round = J;
                    cj |= 1 << 30;                                                    // marks a critical tile
                    break;
                }
            }
        }
        // ---- bulk queue
        if ((int)la_load(&ctrl->bulk_next) < total_bulk) {
            const int t = (int)atomicAdd(&ctrl->bulk_next, 1u);
            if (t < total_bulk) {
                int J = 0, start = 0;
                for (int n; t >= start + (n = la_bulk_count(g, J)); ++J) start += n;
                const int r0q = (J + 1) * g.swq, nq = g.nt - r0q;
                const int sw = (nq < g.swq) ? nq : g.swq, n2 = nq - sw;               // bulk = columns sw .. nq-1
                const int idx = t - start;
                // column c of the n2 x n2 lower triangle starts at c*n2 - c(c-1)/2
                const float bq = 2.0f * n2 + 1.0f;
                int cc = (int)((bq - sqrtf(fmaxf(bq * bq - 8.0f * (float)idx, 0.0f))) * 0.5f);   // estimate, corrected below
                if (cc < 0) cc = 0;
                if (cc > n2 - 1) cc = n2 - 1;
                while (cc > 0 && cc * n2 - cc * (cc - 1) / 2 > idx) --cc;
                while (cc + 1 < n2 && (cc + 1) * n2 - (cc + 1) * cc / 2 <= idx) ++cc;
                const int rr = idx - (cc * n2 - cc * (cc - 1) / 2);
                cj = r0q + sw + cc; ci = cj + rr; round = J;
                break;
            }
        }
        if (c >= total_crit) return make_int3(-2, 0, 0);                              // both queues exhausted
        if (la_load((const unsigned *)&ctrl->err) != 0) return make_int3(-2, 0, 0);
        if (spins < 8) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(100);   // nothing to hand out yet: the chain is the bottleneck
        if (spins > 4000000) { atomicExch(&ctrl->err, -8); return make_int3(-2, 0, 0); }
    }
    bool ok = la_wait_ge<4>(&ctrl->flag[round], 1u, ctrl);                            // L[:, J], d_J final
    ok = ok && la_wait_ge<2>(done + (long)ci * g.nt + (cj & 0xffff), (unsigned)round, ctrl);     // round J-1 of this tile
    return ok ? make_int3(round, ci, cj) : make_int3(-2, 0, 0);
}

__device__ __noinline__ void la_tile(double *K, long ld, const double *dvec, int nbo, int J, int ci, int cj, double *lds) {
    const long C0 = (long)J * nbo;
    GemmArgs g = {};
    g.A = K + C0 * ld; g.lda = ld;
    g.B = K + C0 * ld; g.ldb = ld;
    g.C = K; g.ldc = ld;
    g.K = nbo; g.alpha = -1.0;
    gemm_tile_64<EPI_ACCUM, true, true>(g, lds, (long)ci * SB, (long)cj * SB, dvec + C0);
}
__global__ __launch_bounds__(256, 5) void k_ldlt_workers(double *K, long ld, const double *dvec, int Npad, int nbo, LaCtrl *ctrl,
                                                          unsigned *done, int reserve, ReserveMap rm, int dbg) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * CIP_KT * SB];   // 32 KB
    int *sh = (int *)lds;            // (round, ci, cj) of the next tile: handed over in the first bytes of the GEMM buffer
    if (reserve) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0x7;
        const unsigned cu = (hw >> 8) & 0xf, gse = xcc * 4 + ((hw >> 13) & 0x3);
        if (cu == rm.cu[gse][0] || (reserve > 1 && cu == rm.cu[gse][1])) return;
    }
    LaGeom geo;
    geo.nbo = nbo; geo.nt = Npad / SB; geo.swq = nbo / SB; geo.nrounds = (Npad + nbo - 1) / nbo - 1;
    int total_crit = 0, total_bulk = 0;
    for (int J = 0; J < geo.nrounds; ++J) { total_crit += la_crit_count(geo, J); total_bulk += la_bulk_count(geo, J); }
    int cur_round = -1;
    long busy = 0, ntiles = 0;
    for (;;) {
        if (threadIdx.x == 0) {
            const int3 nx = la_next_tile(geo, total_crit, total_bulk, ctrl, done);
            if (nx.x >= 0 && nx.x != cur_round) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            sh[0] = nx.x; sh[1] = nx.y; sh[2] = nx.z;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int J = sh[0], ci = sh[1], cj = sh[2] & 0xffff;
        const bool critical = (sh[2] >> 30) != 0;
        __syncthreads();                 // everyone has read them: the tile's staging may overwrite the buffer
        if (J < 0) break;
        cur_round = J;
        const long t0 = __builtin_amdgcn_s_memtime();
        // a critical tile out-prioritises the bulk tiles it shares the CU with (MFMA issue is arbitrated by priority)
        if (critical) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);
        if (!(dbg & 64)) la_tile(K, ld, dvec, nbo, J, ci, cj, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(done + (long)ci * geo.nt + cj, (unsigned)(J + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&ctrl->stripdone[cj / geo.swq], 1u);
            busy += __builtin_amdgcn_s_memtime() - t0; ++ntiles;
        }
    }
    if (threadIdx.x == 0) {
        atomicAdd(&ctrl->busy_ticks, (unsigned long long)busy);
        atomicAdd(&ctrl->tiles, (unsigned long long)ntiles);
        atomicAdd(&ctrl->workers_done, 1u);
    }
}

// chain side: publish "block J is final" / wait for the strip of block S
__global__ void k_la_signal(unsigned *flag) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_la_gate(const unsigned *counter, unsigned target, LaCtrl *ctrl) {
    if (threadIdx.x == 0) (void)la_wait_ge<2>(counter, target, ctrl);
}

size_t cip_la_ctrl_bytes(int Npad) { return sizeof(LaCtrl) + sizeof(unsigned) * (size_t)(Npad / SB) * (Npad / SB); }

static int g_ncu = 0;
int cip_la_launch_workers(hipStream_t s, double *K, int Npad, long ld, const double *dvec, int nbo, void *ctrl_dev, int reserve) {
    LaGeom geo;
    geo.nbo = nbo; geo.nt = Npad / SB; geo.swq = nbo / SB; geo.nrounds = (Npad + nbo - 1) / nbo - 1;
    if (geo.nrounds > LA_MAX_ROUNDS) { cip_set_error("look-ahead: too many outer blocks"); return -1; }
    long total = 0;
    for (int J = 0; J < geo.nrounds; ++J) total += la_crit_count(geo, J) + la_bulk_count(geo, J);
    if (!g_ncu) {
        hipDeviceProp_t prop;
        int dev = 0;
        CIP_HIP_CHECK(hipGetDevice(&dev));
        CIP_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        g_ncu = prop.multiProcessorCount;
    }
    if (reserve && reserve_map_init()) return -1;
    LaCtrl *ctrl = (LaCtrl *)ctrl_dev;
    long grid = (long)g_ncu * 5;                             // 32 KB of LDS each: 5 per CU
    if (grid > total) grid = total;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_ldlt_workers, dim3((unsigned)grid), dim3(256), 0, s, K, ld, dvec, Npad, nbo, ctrl,
                       (unsigned *)((char *)ctrl_dev + sizeof(LaCtrl)), reserve, g_rmap,
                       getenv("CIP_LA_DBG") ? atoi(getenv("CIP_LA_DBG")) : 0);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_la_signal(hipStream_t s, void *ctrl_dev, int J) {
    hipLaunchKernelGGL(k_la_signal, dim3(1), dim3(64), 0, s, &((LaCtrl *)ctrl_dev)->flag[J]);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
// wait until every tile of the column strip of outer block S has received the updates of rounds 0 .. S-1
int cip_la_gate(hipStream_t s, void *ctrl_dev, int Npad, int nbo, int S) {
    const int swq = nbo / SB, nq = Npad / SB - S * swq;
    const int sw = nq < swq ? nq : swq;
    const unsigned target = (unsigned)S * (unsigned)(sw * nq - sw * (sw - 1) / 2);
    hipLaunchKernelGGL(k_la_gate, dim3(1), dim3(64), 0, s, &((LaCtrl *)ctrl_dev)->stripdone[S], target, (LaCtrl *)ctrl_dev);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
__global__ void k_la_finish(const LaCtrl *ctrl, int *info) {
    if (threadIdx.x == 0 && ctrl->err != 0) info[3] = ctrl->err;
}
int cip_la_finish(hipStream_t s, void *ctrl_dev, int *info) {
    hipLaunchKernelGGL(k_la_finish, dim3(1), dim3(64), 0, s, (const LaCtrl *)ctrl_dev, info);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}
int cip_la_read_stats(void *ctrl_dev, hipStream_t s, double *busy_ticks, double *tiles, double *workers, int *err) {
    LaCtrl h;
    CIP_HIP_CHECK(hipMemcpyAsync(&h, ctrl_dev, sizeof(LaCtrl), hipMemcpyDeviceToHost, s));
    CIP_HIP_CHECK(hipStreamSynchronize(s));
    if (busy_ticks) *busy_ticks = (double)h.busy_ticks;
    if (tiles) *tiles = (double)h.tiles;
    if (workers) *workers = (double)h.workers_done;
    if (err) *err = h.err;
    return 0;
}

int cip_launch_gemm(hipStream_t s, int epi, const GemmArgs &g) {
    if (g.M <= 0 || g.N <= 0) return 0;
    if (g.M % CIP_NB || g.N % CIP_NB || g.K % CIP_KT || g.K <= 0) {
        cip_set_error("gemm: bad dims M=%d N=%d K=%d", g.M, g.N, g.K);
        return -1;
    }
    const int tm = g.M / CIP_NB, tn = g.N / CIP_NB;
    long tiles;
    if (g.lower) {
        if (g.M != g.N) { cip_set_error("gemm: lower needs M == N"); return -1; }
        tiles = (long)tm * (tm + 1) / 2;
    } else {
        tiles = (long)tm * tn;
    }
    const int by = g.by > 0 ? g.by : 1, bz = g.bz > 0 ? g.bz : 1;
    if (by * bz > 1 || (g.overwrite && epi == EPI_ACCUM && !g.lower)) {
        if (epi != EPI_ACCUM || g.lower) { cip_set_error("gemm: batching needs the plain accumulate form"); return -1; }
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
        if (g.dk) cip_launch_b(k_ldlt_trailing_64<true>, dim3((unsigned)(4 * tiles)), dim3(256), 0, s, g);
        else cip_launch_b(k_ldlt_trailing_64<false>, dim3((unsigned)(4 * tiles)), dim3(256), 0, s, g);
        CIP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (epi == EPI_SYRKQ && g.lower && g_tile == 64) {
        cip_launch_b(k_syrkq_64, dim3((unsigned)(4 * tiles)), dim3(256), 0, s, g);
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
