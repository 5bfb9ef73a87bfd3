// Development probe: what a LONE wave pays per fp64 instruction on gfx950 -- dependent chains against independent streams,
// plain VALU against DPP and v_rcp_f64 (the arithmetic of diag.hip's step A).  Generated text (one kernel per mode), 256 loop trips.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/issue_probe tools/issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k0(long *ticks, double *out, double seed) {
    double x = seed + 1e-3 * threadIdx.x; long t0, t1;
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3e100000\n\t"
        "v_mov_b32 v2, %2\n\tv_mov_b32 v3, %3\n\tv_mov_b32 v4, %2\n\tv_mov_b32 v5, %3\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\tv_mov_b32 v8, %2\n\tv_mov_b32 v9, %3\n\tv_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\t"
        "s_mov_b64 s[22:23], exec\n\ts_movk_i32 s20, 256\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t1:\n\t"
        "v_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\t"
        "s_sub_i32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[22:23]\n\t"
        : "=s"(t0), "=s"(t1) : "v"(__double2loint(x)), "v"(__double2hiint(x))
        : "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v20","v21","v22","v23","s20","s22","s23","memory");
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
    out[threadIdx.x] = x;
}
__global__ void k1(long *ticks, double *out, double seed) {
    double x = seed + 1e-3 * threadIdx.x; long t0, t1;
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3e100000\n\t"
        "v_mov_b32 v2, %2\n\tv_mov_b32 v3, %3\n\tv_mov_b32 v4, %2\n\tv_mov_b32 v5, %3\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\tv_mov_b32 v8, %2\n\tv_mov_b32 v9, %3\n\tv_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\t"
        "s_mov_b64 s[22:23], exec\n\ts_movk_i32 s20, 256\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t1:\n\t"
        "v_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\t"
        "s_sub_i32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[22:23]\n\t"
        : "=s"(t0), "=s"(t1) : "v"(__double2loint(x)), "v"(__double2hiint(x))
        : "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v20","v21","v22","v23","s20","s22","s23","memory");
    if (threadIdx.x == 0) ticks[1] = t1 - t0;
    out[threadIdx.x] = x;
}
__global__ void k2(long *ticks, double *out, double seed) {
    double x = seed + 1e-3 * threadIdx.x; long t0, t1;
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3e100000\n\t"
        "v_mov_b32 v2, %2\n\tv_mov_b32 v3, %3\n\tv_mov_b32 v4, %2\n\tv_mov_b32 v5, %3\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\tv_mov_b32 v8, %2\n\tv_mov_b32 v9, %3\n\tv_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\t"
        "s_mov_b64 s[22:23], exec\n\ts_movk_i32 s20, 256\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t1:\n\t"
        "v_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[6:7], v[6:7], v[20:21], v[22:23]\n\tv_fma_f64 v[8:9], v[8:9], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[6:7], v[6:7], v[20:21], v[22:23]\n\tv_fma_f64 v[8:9], v[8:9], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[6:7], v[6:7], v[20:21], v[22:23]\n\tv_fma_f64 v[8:9], v[8:9], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[6:7], v[6:7], v[20:21], v[22:23]\n\tv_fma_f64 v[8:9], v[8:9], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[6:7], v[6:7], v[20:21], v[22:23]\n\tv_fma_f64 v[8:9], v[8:9], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[6:7], v[6:7], v[20:21], v[22:23]\n\tv_fma_f64 v[8:9], v[8:9], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[6:7], v[6:7], v[20:21], v[22:23]\n\tv_fma_f64 v[8:9], v[8:9], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\tv_fma_f64 v[6:7], v[6:7], v[20:21], v[22:23]\n\tv_fma_f64 v[8:9], v[8:9], v[20:21], v[22:23]\n\t"
        "s_sub_i32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[22:23]\n\t"
        : "=s"(t0), "=s"(t1) : "v"(__double2loint(x)), "v"(__double2hiint(x))
        : "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v20","v21","v22","v23","s20","s22","s23","memory");
    if (threadIdx.x == 0) ticks[2] = t1 - t0;
    out[threadIdx.x] = x;
}
__global__ void k3(long *ticks, double *out, double seed) {
    double x = seed + 1e-3 * threadIdx.x; long t0, t1;
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3e100000\n\t"
        "v_mov_b32 v2, %2\n\tv_mov_b32 v3, %3\n\tv_mov_b32 v4, %2\n\tv_mov_b32 v5, %3\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\tv_mov_b32 v8, %2\n\tv_mov_b32 v9, %3\n\tv_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\t"
        "s_mov_b64 s[22:23], exec\n\ts_movk_i32 s20, 256\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t1:\n\t"
        "v_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\tv_fmac_f64_dpp v[2:3], v[2:3], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
        "s_sub_i32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[22:23]\n\t"
        : "=s"(t0), "=s"(t1) : "v"(__double2loint(x)), "v"(__double2hiint(x))
        : "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v20","v21","v22","v23","s20","s22","s23","memory");
    if (threadIdx.x == 0) ticks[3] = t1 - t0;
    out[threadIdx.x] = x;
}
__global__ void k4(long *ticks, double *out, double seed) {
    double x = seed + 1e-3 * threadIdx.x; long t0, t1;
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3e100000\n\t"
        "v_mov_b32 v2, %2\n\tv_mov_b32 v3, %3\n\tv_mov_b32 v4, %2\n\tv_mov_b32 v5, %3\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\tv_mov_b32 v8, %2\n\tv_mov_b32 v9, %3\n\tv_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\t"
        "s_mov_b64 s[22:23], exec\n\ts_movk_i32 s20, 256\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t1:\n\t"
        "v_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[4:5], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[6:7], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[8:9], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[4:5], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[6:7], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[8:9], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[4:5], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[6:7], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[8:9], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[4:5], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[6:7], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[8:9], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[4:5], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[6:7], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[8:9], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[4:5], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[6:7], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[8:9], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[4:5], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[6:7], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[8:9], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[4:5], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[6:7], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[8:9], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_sub_i32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[22:23]\n\t"
        : "=s"(t0), "=s"(t1) : "v"(__double2loint(x)), "v"(__double2hiint(x))
        : "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v20","v21","v22","v23","s20","s22","s23","memory");
    if (threadIdx.x == 0) ticks[4] = t1 - t0;
    out[threadIdx.x] = x;
}
__global__ void k5(long *ticks, double *out, double seed) {
    double x = seed + 1e-3 * threadIdx.x; long t0, t1;
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3e100000\n\t"
        "v_mov_b32 v2, %2\n\tv_mov_b32 v3, %3\n\tv_mov_b32 v4, %2\n\tv_mov_b32 v5, %3\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\tv_mov_b32 v8, %2\n\tv_mov_b32 v9, %3\n\tv_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\t"
        "s_mov_b64 s[22:23], exec\n\ts_movk_i32 s20, 256\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t1:\n\t"
        "v_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\tv_rcp_f64 v[2:3], v[2:3]\n\t"
        "s_sub_i32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[22:23]\n\t"
        : "=s"(t0), "=s"(t1) : "v"(__double2loint(x)), "v"(__double2hiint(x))
        : "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v20","v21","v22","v23","s20","s22","s23","memory");
    if (threadIdx.x == 0) ticks[5] = t1 - t0;
    out[threadIdx.x] = x;
}
__global__ void k6(long *ticks, double *out, double seed) {
    double x = seed + 1e-3 * threadIdx.x; long t0, t1;
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3e100000\n\t"
        "v_mov_b32 v2, %2\n\tv_mov_b32 v3, %3\n\tv_mov_b32 v4, %2\n\tv_mov_b32 v5, %3\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\tv_mov_b32 v8, %2\n\tv_mov_b32 v9, %3\n\tv_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\t"
        "s_mov_b64 s[22:23], exec\n\ts_movk_i32 s20, 256\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t1:\n\t"
        "v_rcp_f64 v[2:3], v[10:11]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_rcp_f64 v[6:7], v[10:11]\n\tv_rcp_f64 v[8:9], v[10:11]\n\tv_rcp_f64 v[2:3], v[10:11]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_rcp_f64 v[6:7], v[10:11]\n\tv_rcp_f64 v[8:9], v[10:11]\n\tv_rcp_f64 v[2:3], v[10:11]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_rcp_f64 v[6:7], v[10:11]\n\tv_rcp_f64 v[8:9], v[10:11]\n\tv_rcp_f64 v[2:3], v[10:11]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_rcp_f64 v[6:7], v[10:11]\n\tv_rcp_f64 v[8:9], v[10:11]\n\tv_rcp_f64 v[2:3], v[10:11]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_rcp_f64 v[6:7], v[10:11]\n\tv_rcp_f64 v[8:9], v[10:11]\n\tv_rcp_f64 v[2:3], v[10:11]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_rcp_f64 v[6:7], v[10:11]\n\tv_rcp_f64 v[8:9], v[10:11]\n\tv_rcp_f64 v[2:3], v[10:11]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_rcp_f64 v[6:7], v[10:11]\n\tv_rcp_f64 v[8:9], v[10:11]\n\tv_rcp_f64 v[2:3], v[10:11]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_rcp_f64 v[6:7], v[10:11]\n\tv_rcp_f64 v[8:9], v[10:11]\n\t"
        "s_sub_i32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[22:23]\n\t"
        : "=s"(t0), "=s"(t1) : "v"(__double2loint(x)), "v"(__double2hiint(x))
        : "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v20","v21","v22","v23","s20","s22","s23","memory");
    if (threadIdx.x == 0) ticks[6] = t1 - t0;
    out[threadIdx.x] = x;
}
__global__ void k7(long *ticks, double *out, double seed) {
    double x = seed + 1e-3 * threadIdx.x; long t0, t1;
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3e100000\n\t"
        "v_mov_b32 v2, %2\n\tv_mov_b32 v3, %3\n\tv_mov_b32 v4, %2\n\tv_mov_b32 v5, %3\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\tv_mov_b32 v8, %2\n\tv_mov_b32 v9, %3\n\tv_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\t"
        "s_mov_b64 s[22:23], exec\n\ts_movk_i32 s20, 256\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t1:\n\t"
        "v_mov_b64_dpp v[4:5], v[2:3] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\ts_nop 1\n\tv_mov_b64_dpp v[2:3], v[4:5] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\ts_nop 1\n\tv_mov_b64_dpp v[4:5], v[2:3] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\ts_nop 1\n\tv_mov_b64_dpp v[2:3], v[4:5] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\ts_nop 1\n\tv_mov_b64_dpp v[4:5], v[2:3] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\ts_nop 1\n\tv_mov_b64_dpp v[2:3], v[4:5] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\ts_nop 1\n\tv_mov_b64_dpp v[4:5], v[2:3] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\ts_nop 1\n\tv_mov_b64_dpp v[2:3], v[4:5] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\ts_nop 1\n\tv_mov_b64_dpp v[4:5], v[2:3] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\ts_nop 1\n\tv_mov_b64_dpp v[2:3], v[4:5] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\ts_nop 1\n\tv_mov_b64_dpp v[4:5], v[2:3] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\ts_nop 1\n\tv_mov_b64_dpp v[2:3], v[4:5] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\ts_nop 1\n\tv_mov_b64_dpp v[4:5], v[2:3] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\ts_nop 1\n\tv_mov_b64_dpp v[2:3], v[4:5] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\ts_nop 1\n\tv_mov_b64_dpp v[4:5], v[2:3] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[4:5], v[4:5], v[20:21], v[22:23]\n\ts_nop 1\n\tv_mov_b64_dpp v[2:3], v[4:5] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\ts_nop 1\n\t"
        "s_sub_i32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[22:23]\n\t"
        : "=s"(t0), "=s"(t1) : "v"(__double2loint(x)), "v"(__double2hiint(x))
        : "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v20","v21","v22","v23","s20","s22","s23","memory");
    if (threadIdx.x == 0) ticks[7] = t1 - t0;
    out[threadIdx.x] = x;
}
__global__ void k8(long *ticks, double *out, double seed) {
    double x = seed + 1e-3 * threadIdx.x; long t0, t1;
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3e100000\n\t"
        "v_mov_b32 v2, %2\n\tv_mov_b32 v3, %3\n\tv_mov_b32 v4, %2\n\tv_mov_b32 v5, %3\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\tv_mov_b32 v8, %2\n\tv_mov_b32 v9, %3\n\tv_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\t"
        "s_mov_b64 s[22:23], exec\n\ts_movk_i32 s20, 256\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t1:\n\t"
        "v_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp v[2:3], v[10:11], v[22:23] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_sub_i32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[22:23]\n\t"
        : "=s"(t0), "=s"(t1) : "v"(__double2loint(x)), "v"(__double2hiint(x))
        : "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v20","v21","v22","v23","s20","s22","s23","memory");
    if (threadIdx.x == 0) ticks[8] = t1 - t0;
    out[threadIdx.x] = x;
}
__global__ void k9(long *ticks, double *out, double seed) {
    double x = seed + 1e-3 * threadIdx.x; long t0, t1;
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3e100000\n\t"
        "v_mov_b32 v2, %2\n\tv_mov_b32 v3, %3\n\tv_mov_b32 v4, %2\n\tv_mov_b32 v5, %3\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\tv_mov_b32 v8, %2\n\tv_mov_b32 v9, %3\n\tv_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\t"
        "s_mov_b64 s[22:23], exec\n\ts_movk_i32 s20, 256\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t1:\n\t"
        "v_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\tv_mul_f64 v[2:3], v[2:3], v[20:21]\n\t"
        "s_sub_i32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[22:23]\n\t"
        : "=s"(t0), "=s"(t1) : "v"(__double2loint(x)), "v"(__double2hiint(x))
        : "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v20","v21","v22","v23","s20","s22","s23","memory");
    if (threadIdx.x == 0) ticks[9] = t1 - t0;
    out[threadIdx.x] = x;
}
__global__ void k10(long *ticks, double *out, double seed) {
    double x = seed + 1e-3 * threadIdx.x; long t0, t1;
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3e100000\n\t"
        "v_mov_b32 v2, %2\n\tv_mov_b32 v3, %3\n\tv_mov_b32 v4, %2\n\tv_mov_b32 v5, %3\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\tv_mov_b32 v8, %2\n\tv_mov_b32 v9, %3\n\tv_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\t"
        "s_mov_b64 s[22:23], exec\n\ts_movk_i32 s20, 256\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t1:\n\t"
        "v_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[6:7], v[10:11]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[6:7], v[10:11]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[6:7], v[10:11]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[6:7], v[10:11]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[6:7], v[10:11]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[6:7], v[10:11]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[6:7], v[10:11]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[4:5], v[10:11]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_rcp_f64 v[6:7], v[10:11]\n\t"
        "s_sub_i32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[22:23]\n\t"
        : "=s"(t0), "=s"(t1) : "v"(__double2loint(x)), "v"(__double2hiint(x))
        : "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v20","v21","v22","v23","s20","s22","s23","memory");
    if (threadIdx.x == 0) ticks[10] = t1 - t0;
    out[threadIdx.x] = x;
}
__global__ void k11(long *ticks, double *out, double seed) {
    double x = seed + 1e-3 * threadIdx.x; long t0, t1;
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3e100000\n\t"
        "v_mov_b32 v2, %2\n\tv_mov_b32 v3, %3\n\tv_mov_b32 v4, %2\n\tv_mov_b32 v5, %3\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\tv_mov_b32 v8, %2\n\tv_mov_b32 v9, %3\n\tv_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\t"
        "s_mov_b64 s[22:23], exec\n\ts_mov_b64 exec, 0xffff\n\ts_movk_i32 s20, 256\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t1:\n\t"
        "v_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\tv_fma_f64 v[2:3], v[2:3], v[20:21], v[22:23]\n\t"
        "s_sub_i32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[22:23]\n\t"
        : "=s"(t0), "=s"(t1) : "v"(__double2loint(x)), "v"(__double2hiint(x))
        : "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v20","v21","v22","v23","s20","s22","s23","memory");
    if (threadIdx.x == 0) ticks[11] = t1 - t0;
    out[threadIdx.x] = x;
}
__global__ void kmf20(long *ticks, double *out) {
    long t0, t1;
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3e100000\n\t"
        "s_movk_i32 s20, 256\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t1:\n\t"
        "v_mfma_f64_16x16x4_f64 v[24:31], v[20:21], v[22:23], v[24:31]\n\tv_mfma_f64_16x16x4_f64 v[24:31], v[20:21], v[22:23], v[24:31]\n\tv_mfma_f64_16x16x4_f64 v[24:31], v[20:21], v[22:23], v[24:31]\n\tv_mfma_f64_16x16x4_f64 v[24:31], v[20:21], v[22:23], v[24:31]\n\tv_mfma_f64_16x16x4_f64 v[24:31], v[20:21], v[22:23], v[24:31]\n\tv_mfma_f64_16x16x4_f64 v[24:31], v[20:21], v[22:23], v[24:31]\n\tv_mfma_f64_16x16x4_f64 v[24:31], v[20:21], v[22:23], v[24:31]\n\tv_mfma_f64_16x16x4_f64 v[24:31], v[20:21], v[22:23], v[24:31]\n\t"
        "s_sub_i32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_nop 7\n\ts_nop 7\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\t"
        : "=s"(t0), "=s"(t1) :
        : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39",
          "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","s20","memory");
    if (threadIdx.x == 0) ticks[20] = t1 - t0;
    out[threadIdx.x] = 0;
}
__global__ void kmf21(long *ticks, double *out) {
    long t0, t1;
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3e100000\n\t"
        "s_movk_i32 s20, 256\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t1:\n\t"
        "v_mfma_f64_16x16x4_f64 v[24:31], v[20:21], v[22:23], v[24:31]\n\tv_mfma_f64_16x16x4_f64 v[32:39], v[20:21], v[22:23], v[32:39]\n\tv_mfma_f64_16x16x4_f64 v[24:31], v[20:21], v[22:23], v[24:31]\n\tv_mfma_f64_16x16x4_f64 v[32:39], v[20:21], v[22:23], v[32:39]\n\tv_mfma_f64_16x16x4_f64 v[24:31], v[20:21], v[22:23], v[24:31]\n\tv_mfma_f64_16x16x4_f64 v[32:39], v[20:21], v[22:23], v[32:39]\n\tv_mfma_f64_16x16x4_f64 v[24:31], v[20:21], v[22:23], v[24:31]\n\tv_mfma_f64_16x16x4_f64 v[32:39], v[20:21], v[22:23], v[32:39]\n\t"
        "s_sub_i32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_nop 7\n\ts_nop 7\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\t"
        : "=s"(t0), "=s"(t1) :
        : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39",
          "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","s20","memory");
    if (threadIdx.x == 0) ticks[21] = t1 - t0;
    out[threadIdx.x] = 0;
}
__global__ void kmf22(long *ticks, double *out) {
    long t0, t1;
    asm volatile("v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3e100000\n\t"
        "s_movk_i32 s20, 256\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t1:\n\t"
        "v_mfma_f64_16x16x4_f64 v[24:31], v[20:21], v[22:23], v[24:31]\n\tv_mfma_f64_16x16x4_f64 v[32:39], v[20:21], v[22:23], v[32:39]\n\tv_mfma_f64_16x16x4_f64 v[40:47], v[20:21], v[22:23], v[40:47]\n\tv_mfma_f64_16x16x4_f64 v[48:55], v[20:21], v[22:23], v[48:55]\n\tv_mfma_f64_16x16x4_f64 v[24:31], v[20:21], v[22:23], v[24:31]\n\tv_mfma_f64_16x16x4_f64 v[32:39], v[20:21], v[22:23], v[32:39]\n\tv_mfma_f64_16x16x4_f64 v[40:47], v[20:21], v[22:23], v[40:47]\n\tv_mfma_f64_16x16x4_f64 v[48:55], v[20:21], v[22:23], v[48:55]\n\t"
        "s_sub_i32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_nop 7\n\ts_nop 7\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\t"
        : "=s"(t0), "=s"(t1) :
        : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39",
          "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","s20","memory");
    if (threadIdx.x == 0) ticks[22] = t1 - t0;
    out[threadIdx.x] = 0;
}
int main() {
    long *dt; double *dout; hipMalloc(&dt, 64 * 8); hipMalloc(&dout, 64 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        k0<<<1, 64>>>(dt, dout, 1.5);
        k1<<<1, 64>>>(dt, dout, 1.5);
        k2<<<1, 64>>>(dt, dout, 1.5);
        k3<<<1, 64>>>(dt, dout, 1.5);
        k4<<<1, 64>>>(dt, dout, 1.5);
        k5<<<1, 64>>>(dt, dout, 1.5);
        k6<<<1, 64>>>(dt, dout, 1.5);
        k7<<<1, 64>>>(dt, dout, 1.5);
        k8<<<1, 64>>>(dt, dout, 1.5);
        k9<<<1, 64>>>(dt, dout, 1.5);
        k10<<<1, 64>>>(dt, dout, 1.5);
        k11<<<1, 64>>>(dt, dout, 1.5);
        kmf20<<<1, 64>>>(dt, dout); kmf21<<<1, 64>>>(dt, dout); kmf22<<<1, 64>>>(dt, dout);
        hipDeviceSynchronize();
    }
    long h[64]; hipMemcpy(h, dt, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-72s %6.1f clocks per instruction\n", "dependent v_fma_f64", h[0] / (256.0 * 32));
    printf("%-72s %6.1f clocks per instruction\n", "two independent fma chains", h[1] / (256.0 * 32));
    printf("%-72s %6.1f clocks per instruction\n", "four independent fma chains", h[2] / (256.0 * 32));
    printf("%-72s %6.1f clocks per instruction\n", "dependent v_fmac_f64_dpp (DPP source = accumulator) + s_nop 1", h[3] / (256.0 * 32));
    printf("%-72s %6.1f clocks per instruction\n", "four independent v_fmac_f64_dpp", h[4] / (256.0 * 32));
    printf("%-72s %6.1f clocks per instruction\n", "dependent v_rcp_f64", h[5] / (256.0 * 32));
    printf("%-72s %6.1f clocks per instruction\n", "four independent v_rcp_f64", h[6] / (256.0 * 32));
    printf("%-72s %6.1f clocks per instruction\n", "v_mov_b64_dpp -> fma -> (s_nop 1) v_mov_b64_dpp -> fma", h[7] / (256.0 * 32));
    printf("%-72s %6.1f clocks per instruction\n", "v_fmac_f64_dpp dependent through the accumulator only", h[8] / (256.0 * 32));
    printf("%-72s %6.1f clocks per instruction\n", "dependent v_mul_f64", h[9] / (256.0 * 32));
    printf("%-72s %6.1f clocks per instruction\n", "fma chain + independent rcp beside it (alternating)", h[10] / (256.0 * 32));
    printf("%-72s %6.1f clocks per instruction\n", "dependent fma with 16 of 64 lanes (EXEC = one row)", h[11] / (256.0 * 32));
    printf("%-72s %6.1f clocks per MFMA\n", "dependent v_mfma_f64_16x16x4_f64", h[20] / (256.0 * 8));
    printf("%-72s %6.1f clocks per MFMA\n", "two independent MFMA chains", h[21] / (256.0 * 8));
    printf("%-72s %6.1f clocks per MFMA\n", "four independent MFMA chains", h[22] / (256.0 * 8));
    return 0;
}
