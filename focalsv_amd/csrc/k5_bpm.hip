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
// This is integer-ALU bound: ~45 VALU ops per DP column per lane.
#include "fsv_internal.h"

namespace {

struct BpmState {
    uint64_t eq0, eq1, eq2, eq3; // match masks of the y rows inside the band
    uint64_t vp, vn;
};

__device__ __forceinline__ uint64_t pick_eq(const BpmState &s, uint32_t c)
{
    uint64_t lo = (c & 1u) ? s.eq1 : s.eq0;
    uint64_t hi = (c & 1u) ? s.eq3 : s.eq2;
    return (c & 2u) ? hi : lo;
}

// y base at padded-window column j (j = 0 is k bases before the predicted start);
// 4 = outside the read ('N' in the reference's fill_subregion).
__device__ __forceinline__ uint32_t ywin_base(const uint32_t *__restrict__ store, const fsv_wtask &t, int win0, int j)
{
    int p = win0 + j;
    if (p < 0 || p >= t.y_len) return 4u;
    return fsv_base_at(store, t.y_word, t.y_len, t.y_rev, p);
}

__device__ __forceinline__ void eq_set(BpmState &s, uint32_t c, uint64_t bit)
{
    s.eq0 |= (c == 0u) ? bit : 0ull;
    s.eq1 |= (c == 1u) ? bit : 0ull;
    s.eq2 |= (c == 2u) ? bit : 0ull;
    s.eq3 |= (c == 3u) ? bit : 0ull;
}

__global__ __launch_bounds__(256) void k5_bpm_kernel(const uint32_t *__restrict__ store, const fsv_wtask *__restrict__ tasks,
                                                     uint32_t n_tasks, fsv_wres *__restrict__ res)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n_tasks) return;
    const fsv_wtask t = tasks[tid];
    const int n = t.x_len, k = t.k;
    const int wlen = n + 2 * k;
    fsv_wres r;
    r.end_site = -1; r.err = -1; r.y_beg = -1; r.extra_begin = 0; r.extra_end = 0;

    // determine_overlap_region (Correct.cpp:203-250)
    if (t.y_start < 0 || t.y_len <= t.y_start || t.y_len - t.y_start + 2 * k + FSV_K_MAX < wlen) {
        r.extra_begin = -1; r.extra_end = -1;
        res[tid] = r;
        return;
    }
    const int win0 = t.y_start - k; // strand coordinate of padded column 0 (may be negative)
    {
        int ys = win0, olen = min(wlen, t.y_len - ys);
        r.extra_end = (int16_t)(wlen - olen);
        if (ys < 0) { r.extra_begin = (int16_t)(-ys); ys = 0; }
        r.y_beg = ys;
    }

    BpmState s;
    s.eq0 = s.eq1 = s.eq2 = s.eq3 = 0; s.vp = 0; s.vn = 0;
    for (int b = 0; b <= 2 * k; b++) eq_set(s, ywin_base(store, t, win0, b), 1ull << b);

    const uint64_t top = 1ull << (2 * k);
    int err = 0;
    bool dead = false;
    for (int i = 0; i < n; i++) {
        uint32_t c = fsv_base_fwd(store, t.x_word, t.x_start + i);
        uint64_t x = pick_eq(s, c) | s.vn;
        uint64_t d0 = ((s.vp + (x & s.vp)) ^ s.vp) | x;
        uint64_t hn = s.vp & d0;
        uint64_t hp = s.vn | ~(s.vp | d0);
        uint64_t sh = d0 >> 1;
        s.vn = sh & hp;
        s.vp = hn | ~(sh | hp);
        if (!(d0 & 1ull)) {
            ++err;
            if (err - 2 * k > k) { dead = true; break; } // Levenshtein_distance.h:367-375
        }
        if (i + 1 < n) {
            s.eq0 >>= 1; s.eq1 >>= 1; s.eq2 >>= 1; s.eq3 >>= 1;
            eq_set(s, ywin_base(store, t, win0, i + 1 + 2 * k), top);
        }
    }
    if (!dead) {
        // last-column scan, Levenshtein_distance.h:418-457
        int best = -1, site = -1, e = err, ungapped = -1;
        if (e <= k) { best = e; site = n - 1; }
        for (int i = 0; i < 2 * k; ) {
            e += (int)((s.vp >> i) & 1ull);
            e -= (int)((s.vn >> i) & 1ull);
            ++i;
            if (e <= k && (best < 0 || e <= best)) { best = e; site = n - 1 + i; }
            if (i == k) ungapped = e;
        }
        if (best >= 0 && k > 0 && ungapped == best) site = n - 1 + k;
        r.end_site = site; r.err = best;
        if (best < 0) r.end_site = -1;
    }
    res[tid] = r;
}

} // namespace

extern "C" int fsv_bpm_windows_dev(fsv_ctx *ctx, const uint32_t *store_dev, const fsv_wtask *tasks_dev, uint32_t n_tasks,
                                   fsv_wres *res_dev)
{
    if (!ctx || !store_dev || (!tasks_dev && n_tasks) || (!res_dev && n_tasks)) return FSV_EINVAL;
    if (n_tasks == 0) return FSV_OK;
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k5_bpm_kernel, dim3(fsv_grid_for(n_tasks, 256)), dim3(256), 0, ctx->stream, store_dev, tasks_dev,
                       n_tasks, res_dev);
    FSV_HIP(ctx, hipGetLastError());
    return FSV_OK;
}

extern "C" int fsv_bpm_windows(fsv_ctx *ctx, const uint32_t *store, size_t store_words, const fsv_wtask *tasks,
                               uint32_t n_tasks, fsv_wres *res)
{
    if (!ctx || !store || (!tasks && n_tasks) || (!res && n_tasks)) return FSV_EINVAL;
    for (uint32_t i = 0; i < n_tasks; i++)
        if (tasks[i].k > FSV_K_MAX || tasks[i].x_len == 0 || tasks[i].x_len > FSV_WINDOW) return fsv_fail(ctx, FSV_EINVAL, "task k/x_len out of range");
    void *d_store = nullptr, *d_tasks = nullptr, *d_res = nullptr;
    int rc = FSV_OK;
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    if (hipMalloc(&d_store, store_words * 4 + 16) != hipSuccess || hipMalloc(&d_tasks, (size_t)n_tasks * sizeof(fsv_wtask) + 16) != hipSuccess ||
        hipMalloc(&d_res, (size_t)n_tasks * sizeof(fsv_wres) + 16) != hipSuccess) {
        rc = fsv_fail(ctx, FSV_ENOMEM, "hipMalloc failed");
    }
    if (rc == FSV_OK) rc = fsv_h2d(ctx, d_store, store, store_words * 4);
    if (rc == FSV_OK) rc = fsv_h2d(ctx, d_tasks, tasks, (size_t)n_tasks * sizeof(fsv_wtask));
    if (rc == FSV_OK) rc = fsv_bpm_windows_dev(ctx, (const uint32_t *)d_store, (const fsv_wtask *)d_tasks, n_tasks, (fsv_wres *)d_res);
    if (rc == FSV_OK) rc = fsv_d2h(ctx, res, d_res, (size_t)n_tasks * sizeof(fsv_wres));
    (void)hipFree(d_store); (void)hipFree(d_tasks); (void)hipFree(d_res);
    return rc;
}
