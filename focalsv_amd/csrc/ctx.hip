// ctx.hip -- context, error strings, raw memory helpers and the K0 host packer.
#include "fsv_internal.h"
#include <string.h>
#include <atomic>
#include <new>

// contexts alive per device: a context sizes its workspace budget from the device's free memory divided among them
static std::atomic<int> g_live_ctx[64];
int fsv_live_contexts(int device) { return device >= 0 && device < 64 ? g_live_ctx[device].load() : 1; }

extern "C" {

int fsv_version(void) { return 100; }

const char *fsv_strerror(int code)
{
    switch (code) {
    case FSV_OK: return "ok";
    case FSV_ENODEV: return "no usable gfx950 device (this library has no CPU fallback)";
    case FSV_EINVAL: return "invalid argument";
    case FSV_ENOMEM: return "out of memory";
    case FSV_EHIP: return "HIP runtime error";
    case FSV_ECAP: return "caller buffer too small";
    case FSV_EUNSUP: return "unsupported input";
    case FSV_EINTERNAL: return "internal error (an invariant did not hold, or a C++ exception was caught at the boundary)";
    default: return "unknown error";
    }
}

int fsv_ctx_create(int device, fsv_ctx **out)
{
    if (!out) return FSV_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return FSV_ENODEV;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return FSV_ENODEV;
    // the code object only carries gfx950 ISA; refuse anything else up front
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return FSV_ENODEV;
    fsv_ctx *c = new (std::nothrow) fsv_ctx();
    if (!c) return FSV_ENOMEM;
    c->device = device;
    c->n_cu = prop.multiProcessorCount;
    c->clock_khz = prop.clockRate;
    c->hbm_bytes = prop.totalGlobalMem;
    c->name = prop.name;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return FSV_EHIP;
    }
    c->own_stream = true;
    if (device < 64) g_live_ctx[device]++;
    *out = c;
    return FSV_OK;
}

void fsv_ctx_destroy(fsv_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->asm_ws_free) ctx->asm_ws_free(ctx);
    if (ctx->aln_ws_free) ctx->aln_ws_free(ctx);
    if (ctx->own_stream && ctx->stream) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    if (ctx->device >= 0 && ctx->device < 64) g_live_ctx[ctx->device]--;
    delete ctx;
}

int fsv_ctx_set_stream(fsv_ctx *ctx, void *hip_stream)
{
    if (!ctx) return FSV_EINVAL;
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->own_stream && ctx->stream) {
        FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        FSV_HIP(ctx, hipStreamDestroy(ctx->stream));
        ctx->own_stream = false;
        ctx->stream = nullptr;
    }
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        FSV_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return FSV_OK;
}

int fsv_ctx_sync(fsv_ctx *ctx)
{
    if (!ctx) return FSV_EINVAL;
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FSV_OK;
}

const char *fsv_last_error(const fsv_ctx *ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int fsv_device_info(const fsv_ctx *ctx, int *n_cu, int *clock_khz, uint64_t *hbm_bytes, char *name, size_t name_cap)
{
    if (!ctx) return FSV_EINVAL;
    if (n_cu) *n_cu = ctx->n_cu;
    if (clock_khz) *clock_khz = ctx->clock_khz;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    if (name && name_cap) {
        strncpy(name, ctx->name.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    return FSV_OK;
}

int fsv_dev_alloc(fsv_ctx *ctx, size_t bytes, void **dev_ptr)
{
    if (!ctx || !dev_ptr) return FSV_EINVAL;
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    FSV_HIP(ctx, hipMalloc(dev_ptr, bytes ? bytes : 1));
    return FSV_OK;
}

int fsv_dev_free(fsv_ctx *ctx, void *dev_ptr)
{
    if (!ctx) return FSV_EINVAL;
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    FSV_HIP(ctx, hipFree(dev_ptr));
    return FSV_OK;
}

int fsv_h2d(fsv_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes)
{
    if (!ctx) return FSV_EINVAL;
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    FSV_HIP(ctx, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FSV_OK;
}

int fsv_d2h(fsv_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes)
{
    if (!ctx) return FSV_EINVAL;
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    FSV_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FSV_OK;
}

// ---- K0 host packer -------------------------------------------------------------------
size_t fsv_pack_bound(const uint64_t *seq_off, uint32_t n_reads)
{
    size_t w = 0;
    for (uint32_t r = 0; r < n_reads; r++) w += (size_t)((seq_off[r + 1] - seq_off[r] + 15) >> 4);
    return w + 4; // tail slack so that vector loads past the last read stay in bounds
}

int fsv_pack_reads(const char *seqs, const uint64_t *seq_off, uint32_t n_reads, uint32_t *words, size_t words_cap,
                   uint64_t *word_off)
{
    if (!seqs || !seq_off || !words || !word_off) return FSV_EINVAL;
    if (words_cap < fsv_pack_bound(seq_off, n_reads)) return FSV_ECAP;
    uint8_t code[256];
    memset(code, 0, sizeof(code)); // N and anything else -> A
    code[(int)'C'] = code[(int)'c'] = 1;
    code[(int)'G'] = code[(int)'g'] = 2;
    code[(int)'T'] = code[(int)'t'] = 3;
    size_t w = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        word_off[r] = w;
        const char *s = seqs + seq_off[r];
        uint64_t len = seq_off[r + 1] - seq_off[r];
        for (uint64_t i = 0; i < len; i += 16) {
            uint32_t v = 0;
            uint64_t lim = (len - i < 16) ? (len - i) : 16;
            for (uint64_t j = 0; j < lim; j++) v |= (uint32_t)code[(uint8_t)s[i + j]] << (2 * j);
            words[w++] = v;
        }
    }
    word_off[n_reads] = w;
    for (size_t t = w; t < words_cap; t++) words[t] = 0;
    return FSV_OK;
}

} // extern "C"
