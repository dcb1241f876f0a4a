// asm.hip -- host orchestration of the batched per-read-set assembly (fsv_assemble_batch).
//
// Replaces the process boundary `hifiasm -o <prefix> -t T <reads.fa>` + GFA read-back
// (focalsv/3_assembly/run_assembly.py:15-44, post_assembly.py:79-95).  All base-level work runs in the
// kernels of asm_kernels.h; the host only sizes buffers between stages and walks the (tiny, <= a few
// hundred nodes per set) overlap graph, which is host code in hifiasm as well (Overlaps.cpp).
#include "asm_kernels.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <thread>
#include <atomic>

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

// HIP-event timing of individual kernels on the context's stream; resolved once at the end of the batch
struct KTimes {
    struct Rec { int k; hipEvent_t a, b; uint64_t bytes; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    hipEvent_t get() { if (used == pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); pool.push_back(e); } return pool[used++]; }
    void begin(fsv_ctx *ctx, int k, uint64_t bytes) { Rec r{k, get(), get(), bytes}; (void)hipEventRecord(r.a, ctx->stream); recs.push_back(r); }
    void end(fsv_ctx *ctx) { (void)hipEventRecord(recs.back().b, ctx->stream); }
    void reset() { recs.clear(); used = 0; }
    ~KTimes() { for (auto e : pool) (void)hipEventDestroy(e); }
};
enum { KN_SKETCH, KN_UNIQ, KN_CHAIN, KN_BPM, KN_RESCUE, KN_PATH_FAST, KN_PATH_DP, KN_CONSENSUS, KN_REPACK, KN_EXACT, KN_STITCH, KN_COUNT };
const char *const kn_names[KN_COUNT] = {"k_sketch", "k_uniq", "k_chain", "k5_bpm", "k_rescue_accept", "k_path_fast", "k_path_dp", "k_consensus",
                                        "k_repack", "k_exact", "k_stitch"};

struct AsmWs {
    DevBuf store[2], word_off, len, set_start, read_set, pair_base, mz, mz_off, mz_cnt, ovl, tasks, res, paths, counters, dp_list, dp_list2, set_cols, trans, read_flag, changed,
        cols, tmp, gwin_off, gwin_read, sk_ends, sk_low, sk_high, hits, hits_packed, set_hits, ovl_prev, exact_flag, inexact_list, upair_base, upair_tab, ovl_c, gwin_tab, cwin, cwin_len, warn, thr_tab, pieces, contig_out, new_len, unpack_off;
    ChainArgs last_chain;   // arguments of the last k_chain launch (the final pass re-chains a few pairs with another bandwidth)
    uint64_t last_chain_bytes = 0;
    // state of the last run (for fsv_asm_fetch_reads / stats)
    std::vector<uint32_t> h_word_off;
    std::vector<int32_t> h_len;
    const uint32_t *cur_store = nullptr;
    uint32_t n_reads = 0;
    fsv_asm_stats stats;
    KTimes kt;
    void *h_pin = nullptr; size_t h_pin_cap = 0; // pinned host staging (exact hits)
    std::vector<DevBuf *> all()
    {
        return {&store[0], &store[1], &word_off, &len, &set_start, &read_set, &pair_base, &mz, &mz_off, &mz_cnt, &ovl, &tasks, &res, &paths,
                &counters, &dp_list, &dp_list2, &set_cols, &trans, &read_flag, &changed, &cols, &tmp, &gwin_off, &gwin_read, &sk_ends, &sk_low, &sk_high, &hits, &hits_packed, &set_hits, &ovl_prev, &exact_flag, &inexact_list, &upair_base, &upair_tab, &ovl_c, &gwin_tab, &cwin, &cwin_len, &warn, &thr_tab, &pieces, &contig_out, &new_len, &unpack_off};
    }
};

void ws_free(fsv_ctx *ctx)
{
    AsmWs *w = (AsmWs *)ctx->asm_ws;
    if (!w) return;
    for (DevBuf *b : w->all()) if (b->p) (void)hipFree(b->p);
    if (w->h_pin) (void)hipHostFree(w->h_pin);
    delete w;
    ctx->asm_ws = nullptr;
}

AsmWs *ws_get(fsv_ctx *ctx)
{
    if (!ctx->asm_ws) { ctx->asm_ws = new AsmWs(); ctx->asm_ws_free = ws_free; memset(&((AsmWs *)ctx->asm_ws)->stats, 0, sizeof(fsv_asm_stats)); }
    return (AsmWs *)ctx->asm_ws;
}

int ensure(fsv_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap && b.p) return FSV_OK;
    if (b.p) { FSV_HIP(ctx, hipStreamSynchronize(ctx->stream)); FSV_HIP(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    FSV_HIP(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return FSV_OK;
}

#define TRY(x) do { int rc_ = (x); if (rc_ != FSV_OK) return rc_; } while (0)

template <class T> int upload(fsv_ctx *ctx, DevBuf &b, const std::vector<T> &v)
{
    TRY(ensure(ctx, b, v.size() * sizeof(T)));
    FSV_HIP(ctx, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return FSV_OK;
}

struct Timer {
    std::chrono::steady_clock::time_point t0;
    fsv_ctx *ctx;
    explicit Timer(fsv_ctx *c) : ctx(c) { (void)hipStreamSynchronize(c->stream); t0 = std::chrono::steady_clock::now(); }
    double stop() { (void)hipStreamSynchronize(ctx->stream); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

uint8_t thr_for_len_host(int x_len)
{
    // verify_window: threshold = x_len * max_ov_diff_ec (0.04 as a double, truncated), Adjust_Threshold (Correct.h:39)
    if (x_len == FSV_WINDOW) return FSV_K_FULL;
    int t = (int)(x_len * 0.04);
    if (t == 0 && x_len >= 4) t = 1;
    return (uint8_t)t;
}

struct Batch {
    uint32_t n_reads = 0, n_sets = 0, n_pairs = 0, n_upairs = 0;
    std::vector<uint32_t> set_start, read_set, pair_base, upair_base;
};

// per-round geometry derived from the current read lengths
struct Geometry {
    std::vector<uint32_t> word_off, mz_off, gwin_off, gwin_read;
    uint64_t task_bound = 0;
    uint32_t max_words = 1;
};

// minimizer slots of a read: the worst case is one per base (inside a long homopolymer or a short-unit tandem repeat every
// k-mer ties with the window minimum and ha_sketch reports all of them); 16 B x bases is 2 % of HBM for 256 regions
static inline uint64_t mz_slots(int64_t len, int) { return (uint64_t)len + 64; }

int make_geometry(fsv_ctx *ctx, const Batch &B, const std::vector<int32_t> &len, Geometry &G, int mz_w = 51, const std::vector<uint32_t> *fixed_mz_off = nullptr)
{
    G.word_off.assign(B.n_reads + 1, 0); G.mz_off.assign(B.n_reads + 1, 0); G.gwin_off.assign(B.n_reads + 1, 0);
    G.max_words = 1;
    uint64_t w = 0, m = 0, g = 0;
    for (uint32_t r = 0; r < B.n_reads; r++) {
        G.word_off[r] = (uint32_t)w; G.mz_off[r] = (uint32_t)m; G.gwin_off[r] = (uint32_t)g;
        w += (uint64_t)(len[r] + 15) / 16;
        G.max_words = std::max<uint32_t>(G.max_words, (uint32_t)((len[r] + 15) / 16));
        m += mz_slots(len[r], mz_w);
        g += (uint64_t)(len[r] + FSV_WINDOW - 1) / FSV_WINDOW;
    }
    if (w + 4 >= (1ull << 32) || m >= (1ull << 32) || g >= (1ull << 32)) return fsv_fail(ctx, FSV_EUNSUP, "batch too large for 32-bit offsets; split it");
    G.word_off[B.n_reads] = (uint32_t)w; G.mz_off[B.n_reads] = (uint32_t)m; G.gwin_off[B.n_reads] = (uint32_t)g;
    if (fixed_mz_off) G.mz_off = *fixed_mz_off;   // slots that do not move between rounds (sized for the longest a read can get)
    G.gwin_read.resize(g);
    for (uint32_t r = 0; r < B.n_reads; r++) for (uint32_t x = G.gwin_off[r]; x < G.gwin_off[r + 1]; x++) G.gwin_read[x] = r;
    G.task_bound = 0;
    for (uint32_t s = 0; s < B.n_sets; s++) {
        uint64_t nw = 0;
        uint32_t ns = B.set_start[s + 1] - B.set_start[s];
        for (uint32_t r = B.set_start[s]; r < B.set_start[s + 1]; r++) nw += G.gwin_off[r + 1] - G.gwin_off[r];
        if (ns > 1) G.task_bound += nw * (ns - 1);
    }
    return FSV_OK;
}

// sketch + per-read index + chaining on the current store; fills ws.ovl (and ws.tasks when emit_tasks)
int overlap_stage(fsv_ctx *ctx, AsmWs &W, const Batch &B, const Geometry &G, const uint32_t *store, const fsv_asm_params &P, int bw,
                  bool emit_tasks, uint32_t task_cap, const uint32_t *only_changed = nullptr)
{
    Timer ts(ctx);
    TRY(ensure(ctx, W.mz, (size_t)G.mz_off[B.n_reads] * sizeof(fsv_mz)));
    TRY(ensure(ctx, W.mz_cnt, (size_t)B.n_reads * 4));
    TRY(ensure(ctx, W.ovl, (size_t)std::max(1u, B.n_pairs) * sizeof(fsv_ovl)));
    TRY(ensure(ctx, W.ovl_c, (size_t)std::max(1u, B.n_pairs) * sizeof(uint4)));
    TRY(ensure(ctx, W.counters, 64));
    FSV_HIP(ctx, hipMemsetAsync(W.counters.p, 0, 64, ctx->stream));
    W.kt.begin(ctx, KN_SKETCH, (uint64_t)G.word_off[B.n_reads] * 4 + (uint64_t)G.mz_off[B.n_reads] / 4 * sizeof(fsv_mz));
    if (!(only_changed && (P.k & 1))) FSV_HIP(ctx, hipMemsetAsync(W.mz_cnt.p, 0, (size_t)B.n_reads * 4, ctx->stream));
    // (with only_changed the unchanged reads keep their count; the kernel zeroes the others itself)
    if (!(P.k & 1)) only_changed = nullptr; // the replay kernel (even k) always sketches every read
    if (P.k & 1) {
        // position-parallel sketch (odd k): per-read scratch for run ends (4 B / base) and two bit planes, planes zeroed per launch
        const size_t total_words = G.word_off[B.n_reads];
        TRY(ensure(ctx, W.sk_ends, (total_words * 16 + 64) * 4));
        TRY(ensure(ctx, W.sk_low, (total_words + B.n_reads + 8) * 4));
        TRY(ensure(ctx, W.sk_high, (total_words + B.n_reads + 8) * 4));
        FSV_HIP(ctx, hipMemsetAsync(W.sk_low.p, 0, (total_words + B.n_reads + 8) * 4, ctx->stream));
        FSV_HIP(ctx, hipMemsetAsync(W.sk_high.p, 0, (total_words + B.n_reads + 8) * 4, ctx->stream));
        hipLaunchKernelGGL(k_sketch_fast, dim3(B.n_reads), dim3(256), 0, ctx->stream, store, (const uint32_t *)W.word_off.p,
                           (const int32_t *)W.len.p, (const uint32_t *)W.mz_off.p, (fsv_mz *)W.mz.p, (uint32_t *)W.mz_cnt.p, B.n_reads, P.w, P.k,
                           P.hpc, (uint32_t *)W.warn.p, (const uint8_t *)nullptr, (uint32_t *)W.sk_ends.p, (uint32_t *)W.sk_low.p, (uint32_t *)W.sk_high.p,
                           only_changed);
        FSV_HIP(ctx, hipGetLastError());
    } else {
        const uint32_t lds_words = std::min<uint32_t>(G.max_words, 8192u);
        FSV_HIP(ctx, hipFuncSetAttribute((const void *)k_sketch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sketch_lds_bytes(P.w, lds_words)));
        hipLaunchKernelGGL(k_sketch, dim3(B.n_reads), dim3(64), sketch_lds_bytes(P.w, lds_words), ctx->stream, store, (const uint32_t *)W.word_off.p,
                           (const int32_t *)W.len.p, (const uint32_t *)W.mz_off.p, (fsv_mz *)W.mz.p, (uint32_t *)W.mz_cnt.p, B.n_reads, P.w, P.k,
                           P.hpc, (uint32_t *)W.warn.p, (const uint8_t *)nullptr, P.w, lds_words);
        FSV_HIP(ctx, hipGetLastError());
    }
    W.kt.end(ctx);
    // the sort in k_uniq holds a read's minimizers in LDS: size it to the longest list of the batch (16 B per entry), so that
    // short-read batches keep many reads per CU
    std::vector<uint32_t> cnt(B.n_reads);
    FSV_HIP(ctx, hipMemcpyAsync(cnt.data(), W.mz_cnt.p, (size_t)B.n_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t max_raw = 0;
    for (uint32_t r = 0; r < B.n_reads; r++) max_raw = std::max(max_raw, cnt[r]);
    W.kt.begin(ctx, KN_UNIQ, (uint64_t)G.mz_off[B.n_reads] / 4 * sizeof(fsv_mz) * 2);
    if (max_raw <= 1024)
        hipLaunchKernelGGL(k_uniq<1024>, dim3(B.n_reads), dim3(256), 0, ctx->stream, (fsv_mz *)W.mz.p, (const uint32_t *)W.mz_off.p,
                           (uint32_t *)W.mz_cnt.p, (uint32_t *)W.warn.p, only_changed);
    else
        hipLaunchKernelGGL(k_uniq<FSV_UQ_MAX>, dim3(B.n_reads), dim3(256), 0, ctx->stream, (fsv_mz *)W.mz.p, (const uint32_t *)W.mz_off.p,
                           (uint32_t *)W.mz_cnt.p, (uint32_t *)W.warn.p, only_changed);
    FSV_HIP(ctx, hipGetLastError());
    W.kt.end(ctx);
    W.stats.ms_sketch += ts.stop();
    if (B.n_pairs == 0) return FSV_OK;
    Timer tc(ctx);
    ChainArgs A;
    A.store = store; A.word_off = (const uint32_t *)W.word_off.p; A.read_len = (const int32_t *)W.len.p;
    A.set_start = (const uint32_t *)W.set_start.p; A.pair_base = (const uint32_t *)W.pair_base.p; A.upair_base = (const uint32_t *)W.upair_base.p;
    A.mz = (const fsv_mz *)W.mz.p; A.mz_off = (const uint32_t *)W.mz_off.p; A.mz_cnt = (const uint32_t *)W.mz_cnt.p;
    A.ovl = (fsv_ovl *)W.ovl.p; A.tasks = (fsv_wtask *)W.tasks.p; A.task_counter = (uint32_t *)W.counters.p; A.task_cap = task_cap;
    TRY(ensure(ctx, W.set_cols, (size_t)B.n_reads * 4));
    FSV_HIP(ctx, hipMemsetAsync(W.set_cols.p, 0, (size_t)B.n_reads * 4, ctx->stream));
    A.overflow = (uint32_t *)W.counters.p + 1; A.warn = (uint32_t *)W.warn.p; A.set_cols = (uint32_t *)W.set_cols.p; A.thr_tab = (const uint8_t *)W.thr_tab.p;
    A.n_sets = B.n_sets; A.k_score = P.k; A.min_anchors = P.min_anchors; A.min_ovlp = P.min_ovlp; A.bw = bw; A.emit_tasks = emit_tasks ? 1 : 0;
    // algorithmic bytes of the pairwise join: every unordered pair reads both unique-minimizer lists (16 B each) and writes two
    // overlap slots; the window tasks it emits are added once their number is known
    uint64_t chain_bytes = (uint64_t)B.n_pairs * sizeof(fsv_ovl);
    uint32_t max_cnt = 0;
    {
        FSV_HIP(ctx, hipMemcpyAsync(cnt.data(), W.mz_cnt.p, (size_t)B.n_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
        FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (uint32_t s2 = 0; s2 < B.n_sets; s2++) {
            const uint64_t ns = B.set_start[s2 + 1] - B.set_start[s2];
            uint64_t tot = 0;
            for (uint32_t r = B.set_start[s2]; r < B.set_start[s2 + 1]; r++) { tot += cnt[r]; if (ns > 1) max_cnt = std::max(max_cnt, cnt[r]); }
            if (ns > 1) chain_bytes += (ns - 1) * tot * sizeof(fsv_mz); // every unordered pair reads both lists once
        }
    }
    // LDS per pair = 24 B x the longest minimizer list of the batch (rounded up to 64, at most FSV_AMAX): more pairs per CU
    A.upair_tab = (const uint4 *)W.upair_tab.p; A.pair_list = nullptr;
    A.amax = (int32_t)std::min<uint32_t>(FSV_AMAX, std::max<uint32_t>(64u, (max_cnt + 63u) / 64u * 64u));
    W.kt.begin(ctx, KN_CHAIN, chain_bytes);
    hipLaunchKernelGGL(k_chain, dim3(B.n_upairs), dim3(64), (size_t)A.amax * 24, ctx->stream, A);
    FSV_HIP(ctx, hipGetLastError());
    W.kt.end(ctx);
    W.last_chain = A;
    W.last_chain_bytes = chain_bytes;
    W.stats.ms_chain += tc.stop();
    return FSV_OK;
}

// ---- layout (host): containment removal, longest mutual out-arcs, unitig walk ------------------------------
// Oriented node v = 2*read + strand.  A hit (q forward, t on strand rev) with x_e == len(q)-1 and y_s == 0 is the
// arc (q,+) -> (t,rev); its complement is (t,!rev) -> (q,-).  Mirrors ma_hit_contained / ma_hit2arc / asg_arc_del_trans
// on error-free linear data (Overlaps.cpp:1198, 2152, 4531; Overlaps.h:178-246) and ma_ug_seq for the sequence.
struct Piece { uint32_t read, rev, len; };
void layout_set(const int32_t *len, uint32_t n, const fsv_hit *hits, uint32_t n_hit, int min_reads,
                std::vector<std::vector<Piece>> &contigs, bool &fallback)
{
    std::vector<uint8_t> contained(n, 0), used(n, 0);
    std::vector<int32_t> succ(2 * n, -1), sovl(2 * n, 0), pred(2 * n, -1);
    fallback = false;
    // ma_hit2arc (Overlaps.h:178-246) from the query's side of every hit; the mirrored hit supplies the other side.
    // tl5 / tl3 = overhang of the target in front of / behind the overlap on the query's strand (y is strand-corrected).
    struct Geom { int ql, tl, qs, qe, tl5, tl3, ext5, ext3, tspan; bool internal; };
    auto geom = [&](const fsv_hit &h) {
        Geom g;
        g.ql = len[h.q]; g.tl = len[h.t]; g.qs = h.x_s; g.qe = h.x_e + 1; g.tl5 = h.y_s; g.tl3 = g.tl - (h.y_e + 1);
        g.ext5 = std::min(g.qs, g.tl5); g.ext3 = std::min(g.ql - g.qe, g.tl3); g.tspan = h.y_e + 1 - h.y_s;
        g.internal = g.ext5 > 1000 || g.ext3 > 1000 || (g.qe - g.qs) < (g.qe - g.qs + g.ext5 + g.ext3) * 0.8f || g.tspan < (g.tspan + g.ext5 + g.ext3) * 0.8f;
        return g;
    };
    for (uint32_t i = 0; i < n_hit; i++) {
        const fsv_hit &h = hits[i];
        const Geom g = geom(h);
        if (g.internal) continue;
        if (g.qs <= g.tl5 && g.ql - g.qe <= g.tl3) {                       // MA_HT_QCONT
            if (g.qs >= g.tl5 && g.ql - g.qe >= g.tl3) { if (h.q > h.t) contained[h.q] = 1; } // mutual: keep the lower index
            else contained[h.q] = 1;
        }
    }
    for (uint32_t v = 0; v < 2 * n; v++) sovl[v] = 0x7fffffff;
    for (uint32_t i = 0; i < n_hit; i++) {
        const fsv_hit &h = hits[i];
        const Geom g = geom(h);
        if (g.internal || contained[h.q] || contained[h.t]) continue;
        if ((g.qs <= g.tl5 && g.ql - g.qe <= g.tl3) || (g.qs >= g.tl5 && g.ql - g.qe >= g.tl3)) continue; // containments
        if (g.qe - g.qs + g.ext5 + g.ext3 < 50 || g.tspan + g.ext5 + g.ext3 < 50) continue;                 // MA_HT_SHORT_OVLP
        int from, to, l;
        if (g.qs > g.tl5) { from = 2 * (int)h.q; to = 2 * (int)h.t + (int)h.rev; l = g.qs - g.tl5; }          // (q,+) -> (t,rev)
        else { from = 2 * (int)h.q + 1; to = 2 * (int)h.t + (h.rev ? 0 : 1); l = (g.ql - g.qe) - g.tl3; }      // (q,-) -> (t,!rev)
        // every node keeps its nearest successor: the smallest node length = the longest overlap
        if (l < sovl[from] || (l == sovl[from] && succ[from] >= 0 && to < succ[from])) { succ[from] = to; sovl[from] = l; }
    }
    for (uint32_t v = 0; v < 2 * n; v++) { int w = succ[v]; if (w >= 0 && succ[w ^ 1] != (int)(v ^ 1)) succ[v] = -1; }
    for (uint32_t v = 0; v < 2 * n; v++) if (succ[v] >= 0) pred[succ[v]] = (int)v;
    // ma_ug_gen (Overlaps.cpp:7759): vertices in increasing order; the unitig through the first unvisited one is emitted in that
    // vertex's direction, from its start (found by walking the in-arcs back).  So the lowest-numbered read of a chain sits on its
    // forward strand -- the two directions spell reverse complements only while every overlap is exact.
    for (uint32_t v = 0; v < 2 * n; v++) {
        const uint32_t r = v >> 1;
        if (contained[r] || used[r]) continue;
        int start = (int)v, steps = 0;
        while (pred[start] >= 0 && !used[pred[start] >> 1] && steps < (int)(2 * n)) { start = pred[start]; steps++; if (start == (int)v) break; }
        int cnt = 0;
        for (int w = start; w >= 0 && !used[w >> 1] && cnt <= (int)(2 * n); w = succ[w]) { cnt++; if (succ[w] == start) break; }
        if (cnt < min_reads) continue;
        std::vector<Piece> c;
        for (int w = start; w >= 0 && !used[w >> 1]; w = succ[w]) {
            used[w >> 1] = 1;
            const bool more = succ[w] >= 0 && !used[succ[w] >> 1];
            c.push_back(Piece{(uint32_t)(w >> 1), (uint32_t)(w & 1), (uint32_t)(more ? sovl[w] : len[w >> 1])});
        }
        contigs.push_back(std::move(c));
    }
    // no fall-back to a single read: hifiasm's asg_cut_tip (Overlaps.cpp:4666-4709) removes dead-end chains of fewer than four
    // reads, a lone read included, and writes no contig for such a set
    fallback = contigs.empty();
}

} // namespace

extern "C" void fsv_asm_default_params(fsv_asm_params *p)
{
    if (!p) return;
    p->k = 51; p->w = 51; p->hpc = 1; p->n_rounds = 3; p->min_ovlp = 500; p->min_anchors = 3; p->lookback = 64;
    p->bw_ec = 20; p->bw_final = 0; p->min_contig_reads = 4;
}

extern "C" int fsv_assemble_batch_bound(const fsv_readsets *sets, uint64_t *seq_cap, uint32_t *contig_cap)
{
    if (!sets || !sets->read_len) return FSV_EINVAL;
    uint64_t tot = 0;
    for (uint32_t r = 0; r < sets->n_reads; r++) tot += (uint64_t)sets->read_len[r];
    // a contig is a concatenation of prefixes of distinct (corrected) reads; consensus can lengthen a read slightly
    if (seq_cap) *seq_cap = tot + tot / 8 + 1024 * (uint64_t)sets->n_sets + 4096;
    if (contig_cap) *contig_cap = sets->n_reads + sets->n_sets + 1;
    return FSV_OK;
}

// K5 + K6 on caller-supplied tasks: the same kernels fsv_assemble_batch drives, with every task treated as belonging to an
// accepted overlap
extern "C" int fsv_bpm_paths(fsv_ctx *ctx, const uint32_t *store, size_t store_words, const fsv_wtask *tasks, uint32_t n_tasks,
                             fsv_wres *res, fsv_wpath *paths)
{
    if (!ctx || !store || (!tasks && n_tasks) || (!res && n_tasks) || (!paths && n_tasks)) return FSV_EINVAL;
    if (n_tasks == 0) return FSV_OK;
    for (uint32_t i = 0; i < n_tasks; i++)
        if (tasks[i].k > FSV_K_MAX || tasks[i].x_len == 0 || tasks[i].x_len > FSV_WINDOW) return fsv_fail(ctx, FSV_EINVAL, "task k/x_len out of range");
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf d_store, d_tasks, d_res, d_paths, d_ovl, d_list, d_cnt, d_cols, d_cols2;
    auto cleanup = [&]() { for (DevBuf *b : {&d_store, &d_tasks, &d_res, &d_paths, &d_ovl, &d_list, &d_cnt, &d_cols, &d_cols2}) if (b->p) (void)hipFree(b->p); };
    int rc = FSV_OK;
    auto run = [&]() -> int {
        std::vector<fsv_wtask> t(tasks, tasks + n_tasks);
        for (auto &x : t) x.ovl = 0;
        fsv_ovl o; memset(&o, 0, sizeof(o)); o.valid = 1; o.is_match = 1;
        TRY(ensure(ctx, d_store, store_words * 4 + 64));
        FSV_HIP(ctx, hipMemsetAsync(d_store.p, 0, store_words * 4 + 64, ctx->stream));
        FSV_HIP(ctx, hipMemcpyAsync(d_store.p, store, store_words * 4, hipMemcpyHostToDevice, ctx->stream));
        TRY(upload(ctx, d_tasks, t));
        TRY(upload(ctx, d_ovl, std::vector<fsv_ovl>{o}));
        TRY(ensure(ctx, d_res, (size_t)n_tasks * sizeof(fsv_wres)));
        TRY(ensure(ctx, d_paths, (size_t)n_tasks * sizeof(fsv_wpath)));
        TRY(ensure(ctx, d_list, (size_t)n_tasks * 4));
        TRY(ensure(ctx, d_cnt, 64));
        FSV_HIP(ctx, hipMemsetAsync(d_cnt.p, 0, 64, ctx->stream));
        FSV_HIP(ctx, hipMemsetAsync(d_paths.p, 0, (size_t)n_tasks * sizeof(fsv_wpath), ctx->stream));
        TRY(fsv_bpm_windows_dev(ctx, (const uint32_t *)d_store.p, (const fsv_wtask *)d_tasks.p, n_tasks, (fsv_wres *)d_res.p));
        hipLaunchKernelGGL(k_path_fast, dim3(fsv_grid_for(n_tasks, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)d_store.p, (const fsv_ovl *)d_ovl.p,
                           (const fsv_wtask *)d_tasks.p, (const fsv_wres *)d_res.p, n_tasks, (fsv_wpath *)d_paths.p, (uint32_t *)d_list.p,
                           (uint32_t *)d_cnt.p + 2, (uint32_t *)d_cnt.p + 6, true);
        FSV_HIP(ctx, hipGetLastError());
        uint32_t cnt[8];
        FSV_HIP(ctx, hipMemcpyAsync(cnt, d_cnt.p, 32, hipMemcpyDeviceToHost, ctx->stream));
        FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        DevBuf &d_list2 = d_cols2;
        const uint32_t *narrow = (const uint32_t *)d_list.p;
        if (cnt[2]) {
            TRY(ensure(ctx, d_list2, (size_t)cnt[2] * 4));
            hipLaunchKernelGGL(k_path_indel1, dim3(fsv_grid_for(cnt[2], 256)), dim3(256), 0, ctx->stream, (const uint32_t *)d_store.p, (const fsv_wtask *)d_tasks.p,
                               (const fsv_wres *)d_res.p, (const uint32_t *)d_list.p, cnt[2], (fsv_wpath *)d_paths.p, (uint32_t *)d_list2.p, (uint32_t *)d_cnt.p + 7);
            FSV_HIP(ctx, hipGetLastError());
            FSV_HIP(ctx, hipMemcpyAsync(&cnt[2], (uint32_t *)d_cnt.p + 7, 4, hipMemcpyDeviceToHost, ctx->stream));
            FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            narrow = (const uint32_t *)d_list2.p;
        }
        for (int wide = 0; wide < 2; wide++) {
            const uint32_t n_here = wide ? cnt[6] : cnt[2], list0 = wide ? n_tasks - cnt[6] : 0u;
            if (!n_here) continue;
            const uint32_t grid = std::min<uint32_t>(fsv_grid_for(n_here, 64), 8u * (uint32_t)ctx->n_cu);
            TRY(ensure(ctx, d_cols, (size_t)grid * 64 * (FSV_WINDOW + 2) * 3 * (wide ? 8 : 4)));
            if (wide)
                hipLaunchKernelGGL(k_path_dp<uint64_t>, dim3(grid), dim3(64), 0, ctx->stream, (const uint32_t *)d_store.p, (const fsv_wtask *)d_tasks.p,
                                   (const uint32_t *)d_list.p, list0, list0 + n_here, (fsv_wpath *)d_paths.p, (uint64_t *)d_cols.p, grid * 64);
            else
                hipLaunchKernelGGL(k_path_dp<uint32_t>, dim3(grid), dim3(64), 0, ctx->stream, (const uint32_t *)d_store.p, (const fsv_wtask *)d_tasks.p,
                                   narrow, list0, list0 + n_here, (fsv_wpath *)d_paths.p, (uint32_t *)d_cols.p, grid * 64);
            FSV_HIP(ctx, hipGetLastError());
        }
        FSV_HIP(ctx, hipMemcpyAsync(res, d_res.p, (size_t)n_tasks * sizeof(fsv_wres), hipMemcpyDeviceToHost, ctx->stream));
        FSV_HIP(ctx, hipMemcpyAsync(paths, d_paths.p, (size_t)n_tasks * sizeof(fsv_wpath), hipMemcpyDeviceToHost, ctx->stream));
        FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return FSV_OK;
    };
    rc = run();
    cleanup();
    return rc;
}

extern "C" int fsv_asm_last_stats(const fsv_ctx *ctx, fsv_asm_stats *out)
{
    if (!ctx || !out || !ctx->asm_ws) return FSV_EINVAL;
    *out = ((const AsmWs *)ctx->asm_ws)->stats;
    return FSV_OK;
}

extern "C" int fsv_assemble_batch(fsv_ctx *ctx, const fsv_readsets *sets, const fsv_asm_params *params, fsv_contigs *out)
{
    if (!ctx || !sets || !out || !sets->store_dev || !sets->word_off || !sets->read_len || !sets->set_start) return FSV_EINVAL;
    if (!out->seq || !out->off || !out->set || !out->n_reads || !out->set_status) return FSV_EINVAL;
    fsv_asm_params P;
    if (params) P = *params; else fsv_asm_default_params(&P);
    if (P.k < 1 || P.k > 63 || P.w < 1 || P.w > 64 || P.lookback != 64 || P.n_rounds < 0 || P.n_rounds > 16 || P.min_anchors < 1)
        return fsv_fail(ctx, FSV_EINVAL, "fsv_asm_params out of range (k<=63, w<=64, lookback==64)");
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    AsmWs &W = *ws_get(ctx);
    memset(&W.stats, 0, sizeof(W.stats));
    W.kt.reset();
    Timer ttotal(ctx);

    Batch B;
    B.n_reads = sets->n_reads; B.n_sets = sets->n_sets;
    out->n_contigs = 0;
    out->off[0] = 0;
    for (uint32_t s = 0; s < B.n_sets; s++) out->set_status[s] = 0;
    if (B.n_reads == 0 || B.n_sets == 0) return FSV_OK;
    B.set_start.assign(sets->set_start, sets->set_start + B.n_sets + 1);
    if (B.set_start[0] != 0 || B.set_start[B.n_sets] != B.n_reads) return fsv_fail(ctx, FSV_EINVAL, "set_start must span [0, n_reads]");
    B.read_set.resize(B.n_reads); B.pair_base.resize(B.n_sets + 1); B.upair_base.resize(B.n_sets + 1);
    uint64_t np = 0;
    for (uint32_t s = 0; s < B.n_sets; s++) {
        if (B.set_start[s + 1] < B.set_start[s]) return fsv_fail(ctx, FSV_EINVAL, "set_start not monotone");
        uint64_t ns = B.set_start[s + 1] - B.set_start[s];
        B.pair_base[s] = (uint32_t)np;
        B.upair_base[s] = (uint32_t)(np / 2);
        np += ns > 1 ? ns * (ns - 1) : 0;
        for (uint32_t r = B.set_start[s]; r < B.set_start[s + 1]; r++) B.read_set[r] = s;
    }
    if (np >= (1ull << 31)) return fsv_fail(ctx, FSV_EUNSUP, "too many read pairs in one batch; split it");
    B.pair_base[B.n_sets] = (uint32_t)np; B.n_pairs = (uint32_t)np;
    B.upair_base[B.n_sets] = (uint32_t)(np / 2); B.n_upairs = (uint32_t)(np / 2);
    std::vector<int32_t> len(sets->read_len, sets->read_len + B.n_reads);
    for (uint32_t r = 0; r < B.n_reads; r++) if (len[r] < 1 || len[r] >= (1 << 24)) return fsv_fail(ctx, FSV_EUNSUP, "read length must be in [1, 2^24)");

    std::vector<uint8_t> thr(FSV_WINDOW + 1);
    for (int i = 0; i <= FSV_WINDOW; i++) thr[i] = thr_for_len_host(i);
    TRY(upload(ctx, W.thr_tab, thr));
    TRY(upload(ctx, W.set_start, B.set_start));
    TRY(upload(ctx, W.read_set, B.read_set));
    TRY(upload(ctx, W.pair_base, B.pair_base));
    TRY(upload(ctx, W.upair_base, B.upair_base));
    if (B.n_upairs) {
        TRY(ensure(ctx, W.upair_tab, (size_t)B.n_upairs * sizeof(uint4)));
        hipLaunchKernelGGL(k_pair_tab, dim3((B.n_upairs + 255) / 256), dim3(256), 0, ctx->stream, (const uint32_t *)W.set_start.p,
                           (const uint32_t *)W.pair_base.p, (const uint32_t *)W.upair_base.p, B.n_sets, B.n_upairs, (uint4 *)W.upair_tab.p);
        FSV_HIP(ctx, hipGetLastError());
    }
    TRY(ensure(ctx, W.warn, (size_t)B.n_reads * 4));
    FSV_HIP(ctx, hipMemsetAsync(W.warn.p, 0, (size_t)B.n_reads * 4, ctx->stream));
    bool any_unphased = false;
    if (sets->set_flags) {
        std::vector<uint8_t> rf(B.n_reads, 0);
        for (uint32_t s2 = 0; s2 < B.n_sets; s2++)
            if (sets->set_flags[s2] & FSV_SET_UNPHASED) { any_unphased = true; for (uint32_t r = B.set_start[s2]; r < B.set_start[s2 + 1]; r++) rf[r] = 1; }
        if (any_unphased) TRY(upload(ctx, W.read_flag, rf));
    }

    // minimizer slots stay where they are for the whole call (a read grows by a few bases at most when it is corrected): the
    // final pass can then keep the lists of reads the last round did not change
    std::vector<uint32_t> mz_fixed(B.n_reads + 1, 0);
    {
        uint64_t m = 0;
        for (uint32_t r = 0; r < B.n_reads; r++) { mz_fixed[r] = (uint32_t)m; m += mz_slots((int64_t)len[r] + len[r] / 8 + 64, P.w); }
        if (m >= (1ull << 32)) return fsv_fail(ctx, FSV_EUNSUP, "batch too large for 32-bit offsets; split it");
        mz_fixed[B.n_reads] = (uint32_t)m;
    }
    Geometry G;
    TRY(make_geometry(ctx, B, len, G, P.w, &mz_fixed));
    // round 0 reads the caller's store through the caller's word offsets
    std::vector<uint32_t> woff0(B.n_reads + 1);
    for (uint32_t r = 0; r <= B.n_reads; r++) {
        if (sets->word_off[r] >= (1ull << 32)) return fsv_fail(ctx, FSV_EUNSUP, "store larger than 2^32 words; split the batch");
        woff0[r] = (uint32_t)sets->word_off[r];
    }
    G.word_off = woff0;
    const uint32_t *store = sets->store_dev;
    uint64_t reads_in_bytes = 0;
    for (uint32_t r = 0; r < B.n_reads; r++) reads_in_bytes += (uint64_t)(len[r] + 3) / 4;

    for (int round = 0; round < P.n_rounds; round++) {
        TRY(upload(ctx, W.word_off, G.word_off));
        TRY(upload(ctx, W.len, len));
        TRY(upload(ctx, W.mz_off, G.mz_off));
        TRY(upload(ctx, W.gwin_off, G.gwin_off));
        TRY(upload(ctx, W.gwin_read, G.gwin_read));
        if (G.task_bound >= (1ull << 31)) return fsv_fail(ctx, FSV_EUNSUP, "window task bound exceeds 2^31; split the batch");
        const uint32_t task_cap = (uint32_t)std::max<uint64_t>(G.task_bound, 1);
        TRY(ensure(ctx, W.tasks, (size_t)task_cap * sizeof(fsv_wtask)));
        TRY(ensure(ctx, W.res, (size_t)task_cap * sizeof(fsv_wres)));
        TRY(overlap_stage(ctx, W, B, G, store, P, P.bw_ec, true, task_cap));
        uint32_t cnt[4] = {0, 0, 0, 0};
        FSV_HIP(ctx, hipMemcpyAsync(cnt, W.counters.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (cnt[1]) return fsv_fail(ctx, FSV_ECAP, "internal window task buffer overflow");
        const uint32_t n_tasks = cnt[0];
        W.stats.n_pairs += B.n_pairs;
        W.stats.n_windows += n_tasks;
        if (n_tasks) {
            Timer tv(ctx);
            W.kt.begin(ctx, KN_BPM, (uint64_t)n_tasks * (32 + 196 + 16));
            TRY(fsv_bpm_windows_dev(ctx, store, (const fsv_wtask *)W.tasks.p, n_tasks, (fsv_wres *)W.res.p));
            W.kt.end(ctx);
            W.kt.begin(ctx, KN_RESCUE, (uint64_t)n_tasks * 48 + (uint64_t)B.n_pairs * sizeof(fsv_ovl) * 2);
            hipLaunchKernelGGL(k_rescue_accept, dim3(fsv_grid_for(B.n_pairs, 64)), dim3(64), 0, ctx->stream, store, (fsv_ovl *)W.ovl.p,
                               B.n_pairs, (fsv_wtask *)W.tasks.p, (fsv_wres *)W.res.p, (unsigned long long *)((uint32_t *)W.counters.p + 4),
                               (uint4 *)W.ovl_c.p);
            FSV_HIP(ctx, hipGetLastError());
            W.kt.end(ctx);
            W.stats.ms_verify += tv.stop();
            Timer tp(ctx);
            TRY(ensure(ctx, W.paths, (size_t)n_tasks * sizeof(fsv_wpath)));
            TRY(ensure(ctx, W.dp_list, (size_t)n_tasks * 4));
            W.kt.begin(ctx, KN_PATH_FAST, (uint64_t)n_tasks * (48 + 128));
            hipLaunchKernelGGL(k_path_fast, dim3(fsv_grid_for(n_tasks, 256)), dim3(256), 0, ctx->stream, store, (const fsv_ovl *)W.ovl.p,
                               (const fsv_wtask *)W.tasks.p, (const fsv_wres *)W.res.p, n_tasks, (fsv_wpath *)W.paths.p,
                               (uint32_t *)W.dp_list.p, (uint32_t *)W.counters.p + 2, (uint32_t *)W.counters.p + 6, false);
            FSV_HIP(ctx, hipGetLastError());
            W.kt.end(ctx);
            uint32_t cnt2[8];
            FSV_HIP(ctx, hipMemcpyAsync(cnt2, W.counters.p, 32, hipMemcpyDeviceToHost, ctx->stream));
            FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            W.stats.dp_columns += (uint64_t)cnt2[4] | (uint64_t)cnt2[5] << 32;   // rescue re-runs (k_rescue_accept)
            {
                std::vector<uint32_t> sc(B.n_reads);
                FSV_HIP(ctx, hipMemcpyAsync(sc.data(), W.set_cols.p, (size_t)B.n_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
                FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
                for (uint32_t s2 = 0; s2 < B.n_sets; s2++) if (B.set_start[s2] < B.n_reads) W.stats.dp_columns += sc[B.set_start[s2]]; // K5 windows
            }
            uint32_t n_dp = cnt2[2];
            const uint32_t n_dp_wide = cnt2[6];
            const uint32_t *narrow_list = (const uint32_t *)W.dp_list.p;
            if (n_dp) {
                // single-indel windows are settled without the DP (k_path_indel1); the rest is compacted into a second list
                TRY(ensure(ctx, W.dp_list2, (size_t)n_dp * 4));
                uint32_t *n2_dev = (uint32_t *)W.counters.p + 7;
                hipLaunchKernelGGL(k_path_indel1, dim3(fsv_grid_for(n_dp, 256)), dim3(256), 0, ctx->stream, store, (const fsv_wtask *)W.tasks.p,
                                   (const fsv_wres *)W.res.p, (const uint32_t *)W.dp_list.p, n_dp, (fsv_wpath *)W.paths.p, (uint32_t *)W.dp_list2.p, n2_dev);
                FSV_HIP(ctx, hipGetLastError());
                uint32_t n2 = 0;
                FSV_HIP(ctx, hipMemcpyAsync(&n2, n2_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
                FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
                W.stats.n_path_indel1 += n_dp - n2;
                n_dp = n2;
                narrow_list = (const uint32_t *)W.dp_list2.p;
            }
            W.stats.n_path_dp += n_dp + n_dp_wide;
            // narrow bands: [0, n_dp) of the list with 32-bit column words; wide bands: the last n_dp_wide entries with 64-bit words
            for (int wide = 0; wide < 2; wide++) {
                const uint32_t n_here = wide ? n_dp_wide : n_dp, list0 = wide ? n_tasks - n_dp_wide : 0u;
                if (!n_here) continue;
                // persistent grid: as many blocks as the device holds at once (LDS- and wave-limited), each striding through the list
                int per_cu = 0;
                auto launch = [&](auto kern, auto *colp) -> int {
                    FSV_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64, 0));
                    const uint32_t grid = std::min<uint32_t>(fsv_grid_for(n_here, 64), (uint32_t)std::max(1, per_cu) * (uint32_t)ctx->n_cu);
                    const uint32_t stride = grid * 64;
                    TRY(ensure(ctx, W.cols, (size_t)stride * (FSV_WINDOW + 2) * 3 * sizeof(*colp)));
                    W.kt.begin(ctx, KN_PATH_DP, (uint64_t)n_here * (32 + 196 + 128));
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, ctx->stream, store, (const fsv_wtask *)W.tasks.p,
                                       wide ? (const uint32_t *)W.dp_list.p : narrow_list, list0, list0 + n_here, (fsv_wpath *)W.paths.p, (decltype(colp))W.cols.p, stride);
                    return FSV_OK;
                };
                if (wide) TRY(launch(k_path_dp<uint64_t>, (uint64_t *)nullptr));
                else TRY(launch(k_path_dp<uint32_t>, (uint32_t *)nullptr));
                FSV_HIP(ctx, hipGetLastError());
                W.kt.end(ctx);
            }
            W.stats.ms_path += tp.stop();
        }
        // consensus -> corrected windows -> new read store
        Timer tcs(ctx);
        const uint32_t n_gwin = G.gwin_off[B.n_reads];
        TRY(ensure(ctx, W.cwin, (size_t)n_gwin * FSV_CW_STRIDE));
        TRY(ensure(ctx, W.cwin_len, (size_t)n_gwin * 2));
        TRY(ensure(ctx, W.new_len, (size_t)B.n_reads * 4));
        if (!n_tasks) { TRY(ensure(ctx, W.paths, sizeof(fsv_wpath))); FSV_HIP(ctx, hipMemsetAsync(W.ovl_c.p, 0, (size_t)std::max(1u, B.n_pairs) * sizeof(uint4), ctx->stream)); }
        TRY(ensure(ctx, W.gwin_tab, (size_t)std::max(1u, n_gwin) * sizeof(uint4)));
        hipLaunchKernelGGL(k_gwin_tab, dim3(fsv_grid_for(n_gwin, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)W.gwin_read.p, (const uint32_t *)W.gwin_off.p,
                           (const uint32_t *)W.read_set.p, (const uint32_t *)W.set_start.p, (const uint32_t *)W.pair_base.p, n_gwin, (uint4 *)W.gwin_tab.p);
        FSV_HIP(ctx, hipGetLastError());
        ConsArgs C;
        C.store = store; C.word_off = (const uint32_t *)W.word_off.p; C.read_len = (const int32_t *)W.len.p;
        C.read_set = (const uint32_t *)W.read_set.p; C.set_start = (const uint32_t *)W.set_start.p; C.pair_base = (const uint32_t *)W.pair_base.p;
        C.gwin_off = (const uint32_t *)W.gwin_off.p; C.gwin_read = (const uint32_t *)W.gwin_read.p; C.ovl_c = (const uint4 *)W.ovl_c.p;
        C.gwin_tab = (const uint4 *)W.gwin_tab.p; C.tasks = (const fsv_wtask *)W.tasks.p;
        C.paths = (const fsv_wpath *)W.paths.p; C.cwin = (uint8_t *)W.cwin.p; C.cwin_len = (uint16_t *)W.cwin_len.p; C.warn = (uint32_t *)W.warn.p;
        C.n_reads = B.n_reads;
        TRY(ensure(ctx, W.changed, (size_t)B.n_reads * 4));
        FSV_HIP(ctx, hipMemsetAsync(W.changed.p, 0, (size_t)B.n_reads * 4, ctx->stream));
        C.changed = (uint32_t *)W.changed.p;
        if (any_unphased && n_tasks && B.n_pairs) {
            // unphased sets: mark the overlaps that carry the other allele at a heterozygous column, then take them out of
            // the consensus (and, through is_match = 2, out of what the final pass accepts as verified)
            TRY(ensure(ctx, W.trans, (size_t)B.n_pairs * 4));
            FSV_HIP(ctx, hipMemsetAsync(W.trans.p, 0, (size_t)B.n_pairs * 4, ctx->stream));
            hipLaunchKernelGGL(k_het, dim3(n_gwin), dim3(64), 0, ctx->stream, C, n_gwin, (const uint8_t *)W.read_flag.p, (uint32_t *)W.trans.p);
            FSV_HIP(ctx, hipGetLastError());
            hipLaunchKernelGGL(k_apply_trans, dim3(fsv_grid_for(B.n_pairs, 256)), dim3(256), 0, ctx->stream, (fsv_ovl *)W.ovl.p, (uint4 *)W.ovl_c.p,
                               (const uint32_t *)W.trans.p, B.n_pairs);
            FSV_HIP(ctx, hipGetLastError());
        }
        W.kt.begin(ctx, KN_CONSENSUS, (uint64_t)n_tasks * 128 + (uint64_t)n_gwin * (96 + 448));
        hipLaunchKernelGGL(k_consensus, dim3(n_gwin), dim3(64), 0, ctx->stream, C, n_gwin);
        FSV_HIP(ctx, hipGetLastError());
        W.kt.end(ctx);
        hipLaunchKernelGGL(k_newlen, dim3(fsv_grid_for(B.n_reads, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)W.gwin_off.p,
                           (const uint16_t *)W.cwin_len.p, B.n_reads, (int32_t *)W.new_len.p);
        FSV_HIP(ctx, hipGetLastError());
        std::vector<int32_t> nlen(B.n_reads);
        FSV_HIP(ctx, hipMemcpyAsync(nlen.data(), W.new_len.p, (size_t)B.n_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
        FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        Geometry G2;
        TRY(make_geometry(ctx, B, nlen, G2, P.w, &mz_fixed));
        DevBuf &dst = W.store[round & 1];
        const uint32_t total_words = G2.word_off[B.n_reads];
        TRY(ensure(ctx, dst, ((size_t)total_words + 8) * 4));
        // k_repack needs the new offsets/lengths while the old ones are still in use by nothing else: stage them in mz_cnt/new_len
        TRY(upload(ctx, W.unpack_off, G2.word_off));
        W.kt.begin(ctx, KN_REPACK, (uint64_t)n_gwin * 384 + (uint64_t)total_words * 4);
        hipLaunchKernelGGL(k_repack, dim3(fsv_grid_for(total_words, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)W.gwin_off.p,
                           (const uint16_t *)W.cwin_len.p, (const uint8_t *)W.cwin.p, (const uint32_t *)W.unpack_off.p,
                           (const int32_t *)W.new_len.p, B.n_reads, total_words, round + 1 < P.n_rounds ? 1 : 0, (uint32_t *)dst.p);
        FSV_HIP(ctx, hipGetLastError());
        W.kt.end(ctx);
        FSV_HIP(ctx, hipMemsetAsync((uint8_t *)dst.p + (size_t)total_words * 4, 0, 32, ctx->stream));
        W.stats.ms_consensus += tcs.stop();
        store = (const uint32_t *)dst.p;
        len = nlen;
        G = G2;
    }

    // final overlaps on the corrected reads
    Timer tf(ctx);
    auto tr0 = std::chrono::steady_clock::now();
    auto trace = [&](const char *what) { if (getenv("FSV_TRACE")) { (void)hipStreamSynchronize(ctx->stream); auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[fsv] final %-14s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - tr0).count()); tr0 = t; } };
    TRY(upload(ctx, W.word_off, G.word_off));
    TRY(upload(ctx, W.len, len));
    TRY(upload(ctx, W.mz_off, G.mz_off));
    TRY(ensure(ctx, W.tasks, 64));
    // what the last correction round verified (coordinates on the reads as they were before that round) -- the final pass
    // accepts inexact overlaps against it; the slots are about to be overwritten
    const bool have_prev = P.n_rounds > 0 && B.n_pairs > 0;
    if (have_prev) {
        TRY(ensure(ctx, W.ovl_prev, (size_t)B.n_pairs * sizeof(fsv_ovl)));
        FSV_HIP(ctx, hipMemcpyAsync(W.ovl_prev.p, W.ovl.p, (size_t)B.n_pairs * sizeof(fsv_ovl), hipMemcpyDeviceToDevice, ctx->stream));
    }
    // reads the last round left untouched keep that round's minimizer lists (the last round does not reverse-complement)
    TRY(overlap_stage(ctx, W, B, G, store, P, P.bw_final, false, 0, P.n_rounds > 0 ? (const uint32_t *)W.changed.p : nullptr));
    trace("overlaps");
    const fsv_hit *hraw = nullptr;
    std::vector<uint32_t> hit_first(B.n_sets + 1, 0);
    if (B.n_pairs) {
        TRY(ensure(ctx, W.hits, (size_t)B.n_pairs * sizeof(fsv_hit)));
        TRY(ensure(ctx, W.set_hits, (size_t)(2 * B.n_sets + 2) * 4));
        FSV_HIP(ctx, hipMemsetAsync(W.set_hits.p, 0, (size_t)B.n_sets * 4, ctx->stream));
        W.kt.begin(ctx, KN_EXACT, (uint64_t)B.n_pairs * sizeof(fsv_ovl) + W.stats.n_pairs * 0);
        TRY(ensure(ctx, W.exact_flag, (size_t)B.n_upairs + 16));
        hipLaunchKernelGGL(k_exact, dim3(B.n_upairs), dim3(64), 0, ctx->stream, store, (const uint32_t *)W.word_off.p, (const int32_t *)W.len.p,
                           (const uint32_t *)W.read_set.p, (const uint32_t *)W.pair_base.p, (const uint4 *)W.upair_tab.p, (const fsv_ovl *)W.ovl.p,
                           (fsv_hit *)W.hits.p, (uint32_t *)W.set_hits.p, (uint8_t *)W.exact_flag.p);
        FSV_HIP(ctx, hipGetLastError());
        W.kt.end(ctx);
        if (have_prev) {
            // pairs without an exact overlap that the last correction round had verified: gapped re-chain, accept per direction
            TRY(ensure(ctx, W.inexact_list, (size_t)B.n_upairs * 4 + 16));
            uint32_t *n_list_dev = (uint32_t *)W.counters.p + 3;
            FSV_HIP(ctx, hipMemsetAsync(n_list_dev, 0, 4, ctx->stream));
            hipLaunchKernelGGL(k_inexact_list, dim3(fsv_grid_for(B.n_upairs, 256)), dim3(256), 0, ctx->stream, (const uint4 *)W.upair_tab.p,
                               (const uint8_t *)W.exact_flag.p, (const fsv_ovl *)W.ovl_prev.p, B.n_upairs, (uint32_t *)W.inexact_list.p, n_list_dev);
            FSV_HIP(ctx, hipGetLastError());
            uint32_t n_list = 0;
            FSV_HIP(ctx, hipMemcpyAsync(&n_list, n_list_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
            FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (n_list) {
                ChainArgs A2 = W.last_chain;
                A2.bw = 1; A2.emit_tasks = 0; A2.pair_list = (const uint32_t *)W.inexact_list.p;
                // timed like the other k_chain launches (a profiler counts it too); its share of the pair lists as algorithmic bytes
                W.kt.begin(ctx, KN_CHAIN, B.n_upairs ? W.last_chain_bytes / B.n_upairs * n_list : 0);
                hipLaunchKernelGGL(k_chain, dim3(n_list), dim3(64), (size_t)A2.amax * 24, ctx->stream, A2);
                FSV_HIP(ctx, hipGetLastError());
                W.kt.end(ctx);
                hipLaunchKernelGGL(k_accept_inexact, dim3(fsv_grid_for(2 * n_list, 256)), dim3(256), 0, ctx->stream, (const uint4 *)W.upair_tab.p,
                                   (const uint32_t *)W.inexact_list.p, n_list, (const fsv_ovl *)W.ovl.p, (const fsv_ovl *)W.ovl_prev.p,
                                   (const uint32_t *)W.read_set.p, (const uint32_t *)W.pair_base.p, (fsv_hit *)W.hits.p, (uint32_t *)W.set_hits.p);
                FSV_HIP(ctx, hipGetLastError());
            }
            W.stats.n_inexact_candidates = n_list;
        }
        // per-set counts -> offsets; the segments are packed on the device and come back in one copy, already grouped by set.
        // The order inside a set depends on atomics and does not matter: the layout's containment marks and "longest arc,
        // smallest target on ties" choices are order-independent.
        FSV_HIP(ctx, hipMemcpyAsync(hit_first.data() + 1, W.set_hits.p, (size_t)B.n_sets * 4, hipMemcpyDeviceToHost, ctx->stream));
        FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (uint32_t s2 = 0; s2 < B.n_sets; s2++) hit_first[s2 + 1] += hit_first[s2];
        const uint32_t nh = hit_first[B.n_sets];
        if ((size_t)nh * sizeof(fsv_hit) > W.h_pin_cap) {
            if (W.h_pin) FSV_HIP(ctx, hipHostFree(W.h_pin));
            W.h_pin = nullptr; W.h_pin_cap = (size_t)nh * sizeof(fsv_hit) * 5 / 4 + 4096;
            FSV_HIP(ctx, hipHostMalloc(&W.h_pin, W.h_pin_cap, hipHostMallocDefault));
        }
        hraw = (const fsv_hit *)W.h_pin;
        if (nh) {
            uint32_t *first_dev = (uint32_t *)W.set_hits.p + B.n_sets;
            TRY(ensure(ctx, W.hits_packed, (size_t)nh * sizeof(fsv_hit)));
            FSV_HIP(ctx, hipMemcpyAsync(first_dev, hit_first.data(), (size_t)(B.n_sets + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k_hits_compact, dim3(B.n_sets), dim3(256), 0, ctx->stream, (const fsv_hit *)W.hits.p, (const uint32_t *)W.pair_base.p,
                               (const uint32_t *)first_dev, (fsv_hit *)W.hits_packed.p);
            FSV_HIP(ctx, hipGetLastError());
            FSV_HIP(ctx, hipMemcpyAsync(W.h_pin, W.hits_packed.p, (size_t)nh * sizeof(fsv_hit), hipMemcpyDeviceToHost, ctx->stream));
            FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        W.stats.n_exact_overlaps = nh;
    }
    trace("exact+gather");
    std::vector<uint32_t> hwarn(B.n_reads);
    FSV_HIP(ctx, hipMemcpyAsync(hwarn.data(), W.warn.p, (size_t)B.n_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));

    // layout per set (host), then stitch on the device
    std::vector<fsv_piece> pieces;
    uint64_t used = 0;
    uint32_t nc = 0;
    int rc_out = FSV_OK;
    // the sets are independent: lay them out on a few host threads, then emit the contigs in set order
    struct SetLayout { std::vector<std::vector<Piece>> contigs; bool fallback = false; };
    std::vector<SetLayout> lay(B.n_sets);
    {
        const uint32_t nthr = std::max(1u, std::min({8u, std::thread::hardware_concurrency(), B.n_sets / 16 + 1}));
        std::atomic<uint32_t> next{0};
        auto work = [&]() {
            for (uint32_t s = next.fetch_add(1); s < B.n_sets; s = next.fetch_add(1)) {
                const uint32_t r0 = B.set_start[s], ns = B.set_start[s + 1] - r0;
                if (ns == 0) continue;
                const uint32_t nh_s = hit_first[s + 1] - hit_first[s];
                layout_set(len.data() + r0, ns, hraw + hit_first[s], nh_s, P.min_contig_reads, lay[s].contigs, lay[s].fallback);
            }
        };
        std::vector<std::thread> thr;
        for (uint32_t t = 1; t < nthr; t++) thr.emplace_back(work);
        work();
        for (auto &t : thr) t.join();
    }
    for (uint32_t s = 0; s < B.n_sets && rc_out == FSV_OK; s++) {
        const uint32_t r0 = B.set_start[s], ns = B.set_start[s + 1] - r0;
        int32_t st = 0;
        for (uint32_t r = r0; r < r0 + ns; r++) st |= (int32_t)(hwarn[r] & (FSV_W_MZ_TRUNC | FSV_W_ANCHOR_TRUNC | FSV_W_INS_EVENTS | FSV_W_WINDOW_KEPT | FSV_W_INTERNAL));
        if (ns == 0) { out->set_status[s] = st; continue; }
        if (lay[s].fallback) st |= FSV_W_NO_LAYOUT;
        for (auto &c : lay[s].contigs) {
            uint64_t clen = 0;
            for (auto &pc : c) clen += pc.len;
            if (nc >= out->contig_cap || used + clen > out->seq_cap) { rc_out = fsv_fail(ctx, FSV_ECAP, "contig output buffers too small (use fsv_assemble_batch_bound)"); break; }
            for (auto &pc : c) { pieces.push_back(fsv_piece{r0 + pc.read, pc.rev, pc.len, 0u, used}); used += pc.len; }
            out->set[nc] = s; out->n_reads[nc] = (uint32_t)c.size();
            out->off[++nc] = used;
        }
        out->set_status[s] = st;
    }
    trace("layout");
    if (rc_out != FSV_OK) return rc_out;
    out->n_contigs = nc;
    ctx->last_contigs_dev = nullptr;
    ctx->last_contig_off.assign(out->off, out->off + nc + 1);
    if (!pieces.empty()) {
        TRY(upload(ctx, W.pieces, pieces));
        TRY(ensure(ctx, W.contig_out, used + 16));
        W.kt.begin(ctx, KN_STITCH, used * 2);
        hipLaunchKernelGGL(k_stitch, dim3((uint32_t)pieces.size()), dim3(256), 0, ctx->stream, store, (const uint32_t *)W.word_off.p,
                           (const int32_t *)W.len.p, (const fsv_piece *)W.pieces.p, (char *)W.contig_out.p);
        FSV_HIP(ctx, hipGetLastError());
        W.kt.end(ctx);
        ctx->last_contigs_dev = (const char *)W.contig_out.p;
        FSV_HIP(ctx, hipMemcpyAsync(out->seq, W.contig_out.p, used, hipMemcpyDeviceToHost, ctx->stream));
        FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    trace("stitch+d2h");
    W.stats.ms_final += tf.stop();
    W.h_word_off = G.word_off; W.h_len = len; W.cur_store = store; W.n_reads = B.n_reads;
    // algorithmic bytes: 2-bit operands + result of every DP task, reads in once per pass, contigs out
    W.stats.algo_bytes = W.stats.n_windows * 212ull + reads_in_bytes * (uint64_t)(P.n_rounds + 1) + used;
    W.stats.ms_total = ttotal.stop();
    // resolve the per-kernel event timings
    W.stats.n_kernels = KN_COUNT;
    for (int k = 0; k < KN_COUNT; k++) { memset(&W.stats.kernels[k], 0, sizeof(fsv_kernel_stat)); strncpy(W.stats.kernels[k].name, kn_names[k], 23); }
    for (auto &r : W.kt.recs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) W.stats.kernels[r.k].ms += ms;
        W.stats.kernels[r.k].launches++;
        W.stats.kernels[r.k].algo_bytes += r.bytes;
    }
    return FSV_OK;
}

extern "C" int fsv_asm_fetch_reads(fsv_ctx *ctx, char *seq, uint64_t seq_cap, uint64_t *off, uint32_t n_reads)
{
    if (!ctx || !ctx->asm_ws || !seq || !off) return FSV_EINVAL;
    AsmWs &W = *(AsmWs *)ctx->asm_ws;
    if (n_reads != W.n_reads || !W.cur_store) return fsv_fail(ctx, FSV_EINVAL, "no assembled batch with that many reads on this context");
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint64_t> o(n_reads + 1, 0);
    for (uint32_t r = 0; r < n_reads; r++) o[r + 1] = o[r] + (uint64_t)W.h_len[r];
    if (o[n_reads] > seq_cap) return FSV_ECAP;
    DevBuf d_off, d_out;
    int rc = upload(ctx, d_off, o);
    if (rc == FSV_OK) rc = ensure(ctx, d_out, o[n_reads] + 16);
    if (rc == FSV_OK) {
        hipLaunchKernelGGL(k_unpack_reads, dim3(n_reads), dim3(256), 0, ctx->stream, W.cur_store, (const uint32_t *)W.word_off.p,
                           (const int32_t *)W.len.p, (const uint64_t *)d_off.p, (char *)d_out.p);
        if (hipGetLastError() != hipSuccess) rc = FSV_EHIP;
    }
    if (rc == FSV_OK) rc = fsv_d2h(ctx, seq, d_out.p, o[n_reads]);
    if (d_off.p) (void)hipFree(d_off.p);
    if (d_out.p) (void)hipFree(d_out.p);
    memcpy(off, o.data(), (n_reads + 1) * sizeof(uint64_t));
    return rc;
}

extern "C" int fsv_sketch_reads(fsv_ctx *ctx, const fsv_readsets *sets, int32_t w, int32_t k, int32_t hpc, int32_t variant, fsv_mz *out_mz,
                                uint64_t out_cap, uint64_t *out_off)
{
    if (!ctx || !sets || !sets->store_dev || !sets->word_off || !sets->read_len || !out_mz || !out_off) return FSV_EINVAL;
    if (k < 1 || k > 63 || w < 1 || w > 64) return fsv_fail(ctx, FSV_EINVAL, "k <= 63, w <= 64");
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    AsmWs &W = *ws_get(ctx);
    W.kt.reset();
    Batch B;
    B.n_reads = sets->n_reads; B.n_sets = 1; B.n_pairs = 0;
    B.set_start = {0u, B.n_reads};
    out_off[0] = 0;
    if (B.n_reads == 0) return FSV_OK;
    std::vector<int32_t> len(sets->read_len, sets->read_len + B.n_reads);
    for (uint32_t r = 0; r < B.n_reads; r++) if (len[r] < 1 || len[r] >= (1 << 24)) return fsv_fail(ctx, FSV_EUNSUP, "read length must be in [1, 2^24)");
    Geometry G;
    TRY(make_geometry(ctx, B, len, G, w));
    for (uint32_t r = 0; r <= B.n_reads; r++) G.word_off[r] = (uint32_t)sets->word_off[r];
    TRY(upload(ctx, W.word_off, G.word_off));
    TRY(upload(ctx, W.len, len));
    TRY(upload(ctx, W.mz_off, G.mz_off));
    TRY(ensure(ctx, W.warn, (size_t)B.n_reads * 4));
    FSV_HIP(ctx, hipMemsetAsync(W.warn.p, 0, (size_t)B.n_reads * 4, ctx->stream));
    fsv_asm_params P;
    fsv_asm_default_params(&P);
    P.w = w; P.k = (variant == 1) ? (k | 0) : k; P.hpc = hpc;
    // overlap_stage picks the kernel by the parity of k; to force the replay kernel for an odd k, run its launch here
    TRY(ensure(ctx, W.mz, (size_t)G.mz_off[B.n_reads] * sizeof(fsv_mz)));
    TRY(ensure(ctx, W.mz_cnt, (size_t)B.n_reads * 4));
    FSV_HIP(ctx, hipMemsetAsync(W.mz_cnt.p, 0, (size_t)B.n_reads * 4, ctx->stream));
    if ((k & 1) && variant != 1) {
        const size_t total_words = G.word_off[B.n_reads];
        TRY(ensure(ctx, W.sk_ends, (total_words * 16 + 64) * 4));
        TRY(ensure(ctx, W.sk_low, (total_words + B.n_reads + 8) * 4));
        TRY(ensure(ctx, W.sk_high, (total_words + B.n_reads + 8) * 4));
        FSV_HIP(ctx, hipMemsetAsync(W.sk_low.p, 0, (total_words + B.n_reads + 8) * 4, ctx->stream));
        FSV_HIP(ctx, hipMemsetAsync(W.sk_high.p, 0, (total_words + B.n_reads + 8) * 4, ctx->stream));
        hipLaunchKernelGGL(k_sketch_fast, dim3(B.n_reads), dim3(256), 0, ctx->stream, sets->store_dev, (const uint32_t *)W.word_off.p,
                           (const int32_t *)W.len.p, (const uint32_t *)W.mz_off.p, (fsv_mz *)W.mz.p, (uint32_t *)W.mz_cnt.p, B.n_reads, w, k, hpc,
                           (uint32_t *)W.warn.p, (const uint8_t *)nullptr, (uint32_t *)W.sk_ends.p, (uint32_t *)W.sk_low.p, (uint32_t *)W.sk_high.p,
                           (const uint32_t *)nullptr);
    } else {
        const uint32_t lds_words = std::min<uint32_t>(G.max_words, 8192u);
        FSV_HIP(ctx, hipFuncSetAttribute((const void *)k_sketch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sketch_lds_bytes(w, lds_words)));
        hipLaunchKernelGGL(k_sketch, dim3(B.n_reads), dim3(64), sketch_lds_bytes(w, lds_words), ctx->stream, sets->store_dev,
                           (const uint32_t *)W.word_off.p, (const int32_t *)W.len.p, (const uint32_t *)W.mz_off.p, (fsv_mz *)W.mz.p,
                           (uint32_t *)W.mz_cnt.p, B.n_reads, w, k, hpc, (uint32_t *)W.warn.p, (const uint8_t *)nullptr, w, lds_words);
    }
    FSV_HIP(ctx, hipGetLastError());
    std::vector<uint32_t> cnt(B.n_reads);
    FSV_HIP(ctx, hipMemcpyAsync(cnt.data(), W.mz_cnt.p, (size_t)B.n_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t tot = 0;
    for (uint32_t r = 0; r < B.n_reads; r++) {
        const uint32_t c = std::min<uint32_t>(cnt[r], G.mz_off[r + 1] - G.mz_off[r]);
        if (tot + c > out_cap) return fsv_fail(ctx, FSV_ECAP, "out_mz too small");
        FSV_HIP(ctx, hipMemcpyAsync(out_mz + tot, (const fsv_mz *)W.mz.p + G.mz_off[r], (size_t)c * sizeof(fsv_mz), hipMemcpyDeviceToHost, ctx->stream));
        tot += c;
        out_off[r + 1] = tot;
    }
    FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t r = 0; r < B.n_reads; r++)
        std::sort(out_mz + out_off[r], out_mz + out_off[r + 1], [](const fsv_mz &a, const fsv_mz &b) { return a.pos < b.pos; });
    return FSV_OK;
}
