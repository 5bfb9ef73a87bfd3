// Triangular sweeps of the LDL' solve as ONE launch per sweep (HBM-bound: L is read once per sweep).
//
// Role in the reference: the back-substitutions behind `solve3x3` (src/kktsolvers.jl:39-48 `R \ ...`, :259 / :299
// `Z \ rhs`), 2-5 of them per factorisation (predictor, corrector, refinement: src/ConicIP.jl:879, :907, :919).
//
// The first version was 4 dependent gemv launches per 1024-wide block step (32 launches per solve at N = 8192, each
// with one 16-byte load in flight per lane): 2.1 TB/s.  Here a sweep is a single kernel in which every wave owns two
// columns j, j+1 and walks the block row of its columns,
//     forward  (L y = b):    acc_j = sum_{I < J} U[I-block, j] . y_I        (U = L' mirrored into the upper triangle)
//                            r_j   = b_j - acc_j ;   y_j = XT_J[:, j] . r_J   (XT_J = inv(L_JJ)', column j contiguous)
//     backward (L' x = z):   acc_j = sum_{I > J} L[I-block, j] . x_I ;  r_j = z_j - acc_j ;  x_j = X_J[:, j] . r_J
// with the block results handed from workgroup to workgroup through per-block arrival counters instead of kernel
// boundaries:
//   * a block's vector (8 KB) is published with write-through (`sc1`) 16-byte stores, every storing wave drains its
//     stores (`s_waitcnt vmcnt(0)`), the workgroup barriers, one lane adds to the block's counter (agent scope);
//   * consumers poll the counter with an `sc1` load from one lane, then read the vector with `sc1` 16-byte loads
//     (L1-bypassing; MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility");
//   * the matrix operands of the NEXT tile (16 x 16-byte loads per lane) are issued BEFORE the wait, so the hop
//     latency (~2 us) hides behind the HBM stream.
// Logical workgroup order = arrival order (a ticket from an atomic counter), and a workgroup only ever waits for
// lower tickets, so progress never depends on co-residency or on the dispatch order of the hardware.
// Sums are taken in a fixed order: results are bit-reproducible run to run.
#include "cip_internal.h"

#define SOLVE_WG_COLS 32           // columns per workgroup (16 waves x 2): one workgroup per CU at N = 8192, ONE polling lane each

// agent-scope relaxed atomics = `global_load/store_dwordx2 ... sc1` that the compiler schedules and waits for itself (a
// hand-written asm load is invisible to its s_waitcnt / spill logic -- see the GEMM epilogue of the look-ahead workers)
__device__ __forceinline__ v2d ld_sc1(const double *p) {
    return (v2d){__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                 __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
}
__device__ __forceinline__ void st_sc1(double *p, v2d v) {
    __hip_atomic_store(p, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ double wsum64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ONE lane of the workgroup polls (relaxed agent-scope load = `sc1`), the others wait at the barrier -- the first
// version let every wave poll: 4096 pollers on one line, and a sweep took 0.25 ms.  Bounded spin so that a logic error
// can never hang the GPU: after ~0.1 s the lane gives up and raises `*err`.
__device__ __forceinline__ void wait_count(const unsigned *ctr, unsigned target, int *err) {
    if (threadIdx.x == 0) {
        const long t0 = __builtin_amdgcn_s_memtime();
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memtime() - t0 > (1L << 28)) { atomicExch(err, -7); break; }
        }
    }
    __syncthreads();
}

// all of flags[0 .. q] set?  (wave 0 polls, 64 lanes x two flags each, q < 128; the other waves wait at the barrier)
__device__ __forceinline__ void wait_prefix(const unsigned *flags, int q, int *err) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const long t0 = __builtin_amdgcn_s_memtime();
        for (;;) {
            bool ok = true;
            if (lane <= q) ok = __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (lane + 64 <= q) ok = ok && __hip_atomic_load(flags + lane + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memtime() - t0 > (1L << 28)) { if (lane == 0) atomicExch(err, -7); break; }
        }
    }
    __syncthreads();
}

// ctr: [0] ticket, [1 .. nbk] result arrivals per block (in ticket order of the blocks), [1 + nbk ...] one "r published"
// flag per workgroup (ticket order).  A workgroup waits only for workgroups with LOWER tickets: whole earlier blocks
// (their result counter) and, inside its own block, the predecessors whose r entries its triangular diagonal operand
// touches (inv(L_JJ) is lower triangular: y_j needs r_i for i <= j only).
template <int BS, bool FWD>
__global__ __launch_bounds__(1024) void k_ldlt_sweep(const double *__restrict__ K, long ld, const double *__restrict__ Xinv,
                                                        const double *__restrict__ dinv, const double *__restrict__ in,
                                                        double *__restrict__ rbuf, double *__restrict__ out,
                                                        double *__restrict__ out_scaled, unsigned *ctr, int *err, int Npad) {
    constexpr int NC = BS / 128;                       // 128-row chunks per block: one 16-byte load per lane each
    constexpr int WPB = BS / SOLVE_WG_COLS;            // workgroups per block (<= 128)
    __shared__ unsigned s_ticket;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_ticket = atomicAdd(ctr, 1u);
    __syncthreads();
    const int nbk = Npad / BS;
    const int w = (int)s_ticket;                       // logical workgroup index = arrival order
    const int T = w / WPB, q = w - T * WPB;            // block in ticket order, position inside it
    // forward: columns ascending; backward: columns descending (a dependency always points to lower tickets)
    const int j0 = FWD ? w * SOLVE_WG_COLS + wave * 2 : Npad - (w + 1) * SOLVE_WG_COLS + wave * 2;
    const int J = FWD ? T : nbk - 1 - T, jj = j0 - J * BS;
    unsigned *cnt_o = ctr + 1, *rflag = ctr + 1 + nbk;
    const double *col0 = K + (long)j0 * ld, *col1 = col0 + ld;

    double a0 = 0.0, a1 = 0.0;
    v2d m0[NC], m1[NC];
    auto load_tile = [&](int t) {                      // t-th off-diagonal tile of this block row, ticket order
        const long r = (long)(FWD ? t : nbk - 1 - t) * BS + 2 * lane;
#pragma unroll
        for (int c = 0; c < NC; ++c) { m0[c] = *(const v2d *)(col0 + r + 128 * c); m1[c] = *(const v2d *)(col1 + r + 128 * c); }
    };
    if (T > 0) load_tile(0);
    for (int t = 0; t < T; ++t) {
        const long r = (long)(FWD ? t : nbk - 1 - t) * BS + 2 * lane;
        wait_count(cnt_o + t, WPB, err);
        v2d yv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) yv[c] = ld_sc1(out + r + 128 * c);
        wait_vm0();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            a0 = fma(m0[c].x, yv[c].x, a0); a0 = fma(m0[c].y, yv[c].y, a0);
            a1 = fma(m1[c].x, yv[c].x, a1); a1 = fma(m1[c].y, yv[c].y, a1);
        }
        if (t + 1 < T) load_tile(t + 1);               // in flight across the next wait
    }
    // diagonal-block operand: columns jj, jj+1 of the triangular block inverse, only the 128-row chunks that hold
    // non-zeros (forward: rows <= j, backward: rows >= j); independent of every hand-off, issued before the hop
    const double *xb = Xinv + (size_t)J * BS * BS + 2 * lane;
    const int cdiag = jj >> 7;
    v2d x0[NC], x1[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const bool need = FWD ? (c <= cdiag) : (c >= cdiag);
        x0[c] = need ? *(const v2d *)(xb + (long)jj * BS + 128 * c) : (v2d){0.0, 0.0};
        x1[c] = need ? *(const v2d *)(xb + (long)(jj + 1) * BS + 128 * c) : (v2d){0.0, 0.0};
    }
    a0 = wsum64(a0); a1 = wsum64(a1);
    if (lane == 0) {
        const v2d b = *(const v2d *)(in + j0);
        st_sc1(rbuf + j0, (v2d){b.x - a0, b.y - a1});
    }
    wait_vm0();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(rflag + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wait_prefix(rflag + T * WPB, q, err);
    double y0 = 0.0, y1 = 0.0;
    {
        v2d rv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool need = FWD ? (c <= cdiag) : (c >= cdiag);
            rv[c] = need ? ld_sc1(rbuf + (long)J * BS + 2 * lane + 128 * c) : (v2d){0.0, 0.0};
        }
        wait_vm0();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            // entries beyond the diagonal pair belong to higher tickets (not published yet, possibly not even finite):
            // their operand is exactly zero, the product must be too
            const int row = 128 * c + 2 * lane;
            const bool keep = FWD ? (row <= jj) : (row >= jj);
            const v2d rr = keep ? rv[c] : (v2d){0.0, 0.0};
            y0 = fma(x0[c].x, rr.x, y0); y0 = fma(x0[c].y, rr.y, y0);
            y1 = fma(x1[c].x, rr.x, y1); y1 = fma(x1[c].y, rr.y, y1);
        }
    }
    y0 = wsum64(y0); y1 = wsum64(y1);
    if (lane == 0) {
        st_sc1(out + j0, (v2d){y0, y1});
        if (out_scaled) { const v2d d = *(const v2d *)(dinv + j0); *(v2d *)(out_scaled + j0) = (v2d){y0 * d.x, y1 * d.y}; }
    }
    wait_vm0();
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(cnt_o + T, 1u);
}

template <int BS>
static int launch_sweeps(hipStream_t s, const double *K, int Npad, long ld, const LdltWorkspace &ws, double *rhs) {
    const int nbk = Npad / BS;
    const double *X = (BS == CIP_NB) ? ws.Linv : ws.X;
    const double *XT = (BS == CIP_NB) ? ws.LinvT : ws.XT;
    const size_t per = 1 + (size_t)nbk + Npad / SOLVE_WG_COLS;           // ticket, block counters, per-workgroup flags
    unsigned *cf = ws.sweep_ctr, *cb = ws.sweep_ctr + per;
    CIP_HIP_CHECK(hipMemsetAsync(ws.sweep_ctr, 0, sizeof(unsigned) * 2 * per, s));
    const dim3 grid(Npad / SOLVE_WG_COLS), block(1024);
    // forward: b = rhs -> y (ws.ybuf), z = D^-1 y (ws.zbuf); backward: z -> x (rhs)
    hipLaunchKernelGGL((k_ldlt_sweep<BS, true>), grid, block, 0, s, K, ld, XT, ws.dinv, rhs, ws.tmp, ws.ybuf, ws.zbuf, cf, ws.info + 1, Npad);
    hipLaunchKernelGGL((k_ldlt_sweep<BS, false>), grid, block, 0, s, K, ld, X, ws.dinv, ws.zbuf, ws.tmp, rhs, (double *)nullptr, cb, ws.info + 1, Npad);
    CIP_HIP_CHECK(hipGetLastError());
    return 0;
}

int cip_ldlt_solve_sweeps(hipStream_t s, const double *K, int Npad, long ld, const LdltWorkspace &ws, double *rhs) {
    switch (ws.Bs) {
        case 1024: return launch_sweeps<1024>(s, K, Npad, ld, ws, rhs);
        case 512: return launch_sweeps<512>(s, K, Npad, ld, ws, rhs);
        case 256: return launch_sweeps<256>(s, K, Npad, ld, ws, rhs);
        default: return launch_sweeps<128>(s, K, Npad, ld, ws, rhs);
    }
}
