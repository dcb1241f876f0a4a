// fsv_internal.h -- shared by the HIP translation units of libfocalsv_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>
#include "focalsv_hip.h"

struct fsv_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cu = 0;
    int clock_khz = 0;
    uint64_t hbm_bytes = 0;
    std::string name;
    std::string last_error;
    void *asm_ws = nullptr;                 // assembly workspace (asm.hip)
    void (*asm_ws_free)(fsv_ctx *) = nullptr;
    // contigs of the last fsv_assemble_batch, still on the device (ASCII, back to back) for a device-resident hand-off to the aligner
    const char *last_contigs_dev = nullptr;
    std::vector<uint64_t> last_contig_off;
    void *aln_ws = nullptr;                 // alignment workspace (aln.hip)
    void (*aln_ws_free)(fsv_ctx *) = nullptr;
};

#define FSV_HIP(ctx, call)                                                              \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);      \
            return (e_ == hipErrorOutOfMemory) ? FSV_ENOMEM : FSV_EHIP;                 \
        }                                                                               \
    } while (0)

static inline int fsv_fail(fsv_ctx *ctx, int code, const char *msg)
{
    if (ctx) ctx->last_error = msg;
    return code;
}

// ---- device helpers -----------------------------------------------------------------
// 2-bit read store: 16 bases per uint32 word (see focalsv_hip.h).
__device__ __forceinline__ uint32_t fsv_base_fwd(const uint32_t *__restrict__ store, uint32_t word_off, int pos)
{
    return (store[word_off + ((uint32_t)pos >> 4)] >> (((uint32_t)pos & 15u) << 1)) & 3u;
}

// base at strand coordinate p of a read of length len; rev = reverse complement strand.
__device__ __forceinline__ uint32_t fsv_base_at(const uint32_t *__restrict__ store, uint32_t word_off, int len, int rev, int p)
{
    int q = rev ? (len - 1 - p) : p;
    uint32_t b = fsv_base_fwd(store, word_off, q);
    return rev ? (3u - b) : b;
}

int fsv_live_contexts(int device);   // ctx.hip: contexts alive on a device

// the C entry points never let a C++ exception through (a std::bad_alloc from a vector sized by caller data, a std::system_error
// from a thread that could not start, would otherwise terminate a Python process that came in through ctypes)
#define FSV_GUARD(ctx, call)                                                            \
    try { return (call); }                                                              \
    catch (const std::bad_alloc &) { return fsv_fail(ctx, FSV_ENOMEM, "out of host memory"); } \
    catch (const std::exception &e) { if (ctx) (ctx)->last_error = std::string("exception: ") + e.what(); return FSV_EINTERNAL; } \
    catch (...) { return fsv_fail(ctx, FSV_EINTERNAL, "unknown exception"); }

static inline unsigned fsv_grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

// K5 with the task count left on the device (k5_bpm.hip): n_tasks sizes the grid, *n_dev is the count the kernel uses
// k_cap: the largest threshold of the batch's error model (31 = hifiasm's; above it every window goes through the wide-band kernel)
int fsv_bpm_windows_dev_n(fsv_ctx *ctx, const uint32_t *store_dev, const fsv_wtask *tasks_dev, uint32_t n_tasks, const uint32_t *n_dev,
                          fsv_wres *res_dev, int k_cap);
