// k5_bpm.hip -- batched banded bit-parallel edit distance (hifiasm K5) for gfx950.
//
// Replaces Reserve_Banded_BPM / Reserve_Banded_BPM_4_SSE_only as driven by
// verify_window (hifiasm-0.14 Levenshtein_distance.h:274-461, 893-1198;
// Correct.cpp:203-250, 306-531).
//
// Mapping.  The reference packs 4 windows into one SSE register; here one
// 64-lane wavefront carries 64 independent windows, one per lane.  The whole DP
// state of a window (4 match masks, VP, VN; 64-bit words as in the scalar
// reference so that bits above the band behave identically) lives in 12 VGPRs;
// nothing is staged in LDS and nothing is written back until the final
// (end_site, err).  Operands are read from the 2-bit store: 94 + 102 bytes per
// full window instead of the 375 + 405 ASCII bytes the CPU code unpacks
// (recover_UC_Read_sub_region, Process_Read.cpp:608, 23 % of the CPU profile).
// This is integer-ALU bound: ~30 VALU ops per DP column per lane in the 32-bit form (bpm_run32: bands up to 31 rows,
// every first-pass window), ~50 in the 64-bit one.
#include "bpm_device.h"
#include <algorithm>

namespace {

__global__ __launch_bounds__(256) void k5_bpm_kernel(const uint32_t *__restrict__ store, const fsv_wtask *__restrict__ tasks,
                                                     uint32_t n_tasks, fsv_wres *__restrict__ res, const uint32_t *__restrict__ n_dev)
{
    if (n_dev) n_tasks = min(*n_dev, n_tasks);   // the grid covers the task bound; the count stays on the device (and is clamped to the bound: k_chain leaves it above the bound on overflow)
    uint32_t blk;
    if (!xcd_block((n_tasks + 255u) >> 8, blk)) return;
    const uint32_t tid = blk * blockDim.x + threadIdx.x;
    if (tid >= n_tasks) return;
    const fsv_wtask t = tasks[tid];
    fsv_wres r;
    // Shortcut for a window that matches exactly on its predicted diagonal (nearly every window from the second correction
    // round on): the DP then reports distance 0, and its end-site rule (Levenshtein_distance.h:418-457: the last site with the
    // minimum, overridden by the ungapped site when that attains it) picks the diagonal's end n-1+k.  A 24-word XOR compare
    // instead of 375 DP columns; lanes that differ anywhere run the DP as before.
    bool exact = bpm_window_geometry(t, r);
    if (exact) {
        const int n = t.x_len;
        uint32_t acc = 0;
        // 64 bases a fetch (one dwordx4 + one dword per read): a quarter of the load instructions of the 16-base compare
#pragma unroll
        for (int c = 0; c < (FSV_WINDOW + 63) / 64; c++) {
            if (c * 64 < n) {
                uint32_t xb[4], yb[4], yv[4];
                fetch64_x(store, t.x_word, t.x_start + c * 64, xb);
                fetch64(store, t.y_word, t.y_len, t.y_rev, t.y_start + c * 64, yb, yv);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int i = c * 64 + j * 16;
                    if (i < n) {
                        const int lim = min(16, n - i);
                        const uint32_t fm = lim < 16 ? (1u << (2 * lim)) - 1u : 0xffffffffu, vm = (1u << lim) - 1u;
                        acc |= ((xb[j] ^ yb[j]) & fm) | ((yv[j] & vm) ^ vm);
                    }
                }
            }
        }
        exact = acc == 0u;
        if (exact) { r.end_site = n - 1 + t.k; r.err = 0; }
    }
    if (!exact) {
        // bands of at most 31 rows (k <= 15: every first-pass window) run the recurrence in 32-bit words
        if (t.k <= 15) { BpmNoSink32 none; bpm_run32(store, t, r, none); }
        else bpm_run(store, t, r, BpmNoSink());
    }
    res[tid] = r;
}

// K5 for wide bands (k up to FSV_K_WIDE = 95, 191 rows in six 32-bit limbs): every window of a batch whose error model allows
// thresholds above 31 (fsv_asm_params.k_cap > 31: the ONT profile) goes through this kernel, whatever its own k
__global__ __launch_bounds__(256) void k5_bpm_wide_kernel(const uint32_t *__restrict__ store, const fsv_wtask *__restrict__ tasks,
                                                          uint32_t n_tasks, fsv_wres *__restrict__ res, const uint32_t *__restrict__ n_dev, int k_cap)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n_tasks = min(*n_dev, n_tasks);
    if (tid >= n_tasks) return;
    const fsv_wtask t = tasks[tid];
    fsv_wres r;
    WideNoSink none;
    bpm_run_wide(store, t, r, none, k_cap);
    res[tid] = r;
}

} // namespace

extern "C" int fsv_bpm_windows_dev(fsv_ctx *ctx, const uint32_t *store_dev, const fsv_wtask *tasks_dev, uint32_t n_tasks,
                                   fsv_wres *res_dev)
{
    return fsv_bpm_windows_dev_n(ctx, store_dev, tasks_dev, n_tasks, nullptr, res_dev, FSV_K_MAX);
}

// n_dev != NULL: n_tasks is only the bound the grid is sized for; the kernel reads the actual count from the device
int fsv_bpm_windows_dev_n(fsv_ctx *ctx, const uint32_t *store_dev, const fsv_wtask *tasks_dev, uint32_t n_tasks, const uint32_t *n_dev,
                          fsv_wres *res_dev, int k_cap)
{
    if (!ctx || !store_dev || (!tasks_dev && n_tasks) || (!res_dev && n_tasks)) return FSV_EINVAL;
    if (n_tasks == 0) return FSV_OK;
    if (k_cap > FSV_K_WIDE) return fsv_fail(ctx, FSV_EUNSUP, "window thresholds above 95 (bands above 191 rows) are not built");
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    if (k_cap > FSV_K_MAX)
        hipLaunchKernelGGL(k5_bpm_wide_kernel, dim3(fsv_grid_for(n_tasks, 256)), dim3(256), 0, ctx->stream, store_dev, tasks_dev,
                           n_tasks, res_dev, n_dev, k_cap);
    else
    hipLaunchKernelGGL(k5_bpm_kernel, dim3((fsv_grid_for(n_tasks, 256) + 7u) & ~7u), dim3(256), 0, ctx->stream, store_dev, tasks_dev,
                       n_tasks, res_dev, n_dev);
    FSV_HIP(ctx, hipGetLastError());
    return FSV_OK;
}

extern "C" int fsv_bpm_windows(fsv_ctx *ctx, const uint32_t *store, size_t store_words, const fsv_wtask *tasks,
                               uint32_t n_tasks, fsv_wres *res)
{
    if (!ctx || !store || (!tasks && n_tasks) || (!res && n_tasks)) return FSV_EINVAL;
    int kmax = 0;
    for (uint32_t i = 0; i < n_tasks; i++) {
        if (tasks[i].k > FSV_K_WIDE || tasks[i].x_len == 0 || tasks[i].x_len > FSV_WINDOW) return fsv_fail(ctx, FSV_EINVAL, "task k/x_len out of range");
        kmax = std::max<int>(kmax, tasks[i].k);
    }
    const int k_cap = kmax > FSV_K_MAX ? kmax : FSV_K_MAX;     // a threshold above 31 anywhere: the whole list through the wide kernel
    void *d_store = nullptr, *d_tasks = nullptr, *d_res = nullptr;
    int rc = FSV_OK;
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    if (hipMalloc(&d_store, store_words * 4 + 16) != hipSuccess || hipMalloc(&d_tasks, (size_t)n_tasks * sizeof(fsv_wtask) + 16) != hipSuccess ||
        hipMalloc(&d_res, (size_t)n_tasks * sizeof(fsv_wres) + 16) != hipSuccess) {
        rc = fsv_fail(ctx, FSV_ENOMEM, "hipMalloc failed");
    }
    if (rc == FSV_OK) rc = fsv_h2d(ctx, d_store, store, store_words * 4);
    if (rc == FSV_OK) rc = fsv_h2d(ctx, d_tasks, tasks, (size_t)n_tasks * sizeof(fsv_wtask));
    if (rc == FSV_OK) rc = fsv_bpm_windows_dev_n(ctx, (const uint32_t *)d_store, (const fsv_wtask *)d_tasks, n_tasks, nullptr, (fsv_wres *)d_res, k_cap);
    if (rc == FSV_OK) rc = fsv_d2h(ctx, res, d_res, (size_t)n_tasks * sizeof(fsv_wres));
    (void)hipFree(d_store); (void)hipFree(d_tasks); (void)hipFree(d_res);
    return rc;
}
