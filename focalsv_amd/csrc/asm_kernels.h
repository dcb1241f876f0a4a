// asm_kernels.h -- device kernels of the per-read-set assembly (gfx950).  Included once by asm.hip.
//
// Stage map (reference = hifiasm-0.14 under software/, restated in oracle/asm.c which these kernels
// must match bit for bit):
//   k_sketch          ha_sketch                                  sketch.cpp:39-137
//   k_uniq            per-read index (hash occurs once)          htab.cpp:917-998 restated
//   k_chain           anchors, chain_DP, extension, window list  anchor.cpp:60-178; Hash_Table.cpp:425-616, 83-243; Correct.cpp:306-531
//   k5 (bpm_device.h) Reserve_Banded_BPM                         Levenshtein_distance.h:274-461
//   k_rescue_accept   recalcate_window_advance (right pass), 0.9 / 0.03 filters   Correct.cpp:2629-3023, 725
//   k_path_fast/_dp   try_cigar, Reserve_Banded_BPM_PATH, generate_cigar          Levenshtein_distance.h:465-888; Correct.cpp:1302-1536
//   k_consensus       window_consensus / get_seq_from_Graph as a column vote       Correct.cpp:4010-4195
//   k_newlen/k_repack worker_ec_save (+ reverse complement)      Assembly.cpp:706-767
//   k_exact           if_exact_match                             Assembly.cpp:894-974
//   k_stitch          ma_ug_seq                                  Overlaps.cpp:8969-9034
#pragma once
#include "bpm_device.h"
#include <type_traits>

#define FSV_AMAX       1024  // anchors per read pair held in LDS
#define FSV_AMAX_WIDE_LONG 2560  // ... the same second pass in the long layout (24 B per anchor: 60 KB, a workgroup's LDS limit without opting in to more)
#define FSV_AMAX_WIDE  4096  // ... for ONT-profile batches: k = 15 minimizers every ~8 bases, corrected reads share all of them (a 25 kb overlap: ~3 000)
#define FSV_UQ_MAX     4096  // minimizers per read sorted in LDS
#define FSV_PATH_CAP    416  // ops per window path: x_len (<= 375) + y-only ops (<= k <= 31)
#define FSV_CW_STRIDE   448  // bytes reserved per corrected grid window
#define FSV_INS_MAXLEN   12
#define FSV_SB_MAXERR    7   // k_path_sb: distances it holds in one word per column (2 x 7 + 1 rows x 2 bits)
#define FSV_SB_QUADS ((FSV_WINDOW + 3) / 4)
#define FSV_FR_MAXERR    3   // k_path_fr: distances it walks without the DP matrix
#define FSV_EV_CAP_WIDE 2048 // ... for ONT-profile batches (wide bands): ~25 inserted-base events per overlap and window
#define FSV_EV_CAP     256   // insertion events per grid window (HiFi at 30x: ~8; more sets the read's warning bit 8 and drops the excess)

// fsv_wpath (include/focalsv_hip.h): 128 bytes per window task; state 2 = queued for the DP kernel (internal)
static_assert(sizeof(fsv_wpath) == 128, "fsv_wpath layout");

namespace { // every translation unit that includes this header gets its own copy of the kernels

// ------------------------------------------------------------------------------------------------ k_sketch
__device__ __forceinline__ uint64_t mix64(uint64_t key)
{
    key = ~key + (key << 21);
    key = key ^ key >> 24;
    key = (key + (key << 3)) + (key << 8);
    key = key ^ key >> 14;
    key = (key + (key << 2)) + (key << 4);
    key = key ^ key >> 28;
    key = key + (key << 31);
    return key;
}

// One wavefront per read.  The minimizer recurrence is sequential, but what it emits at a position only depends on
// the w entries around it (and on the k HPC bases behind them), so every lane replays the recurrence over its own 1/64
// of the read plus a warm-up of w+k+4 homopolymer runs in front and w+2 runs behind, and keeps only the minimizers whose
// end position falls inside its own slice.  Output order is arbitrary (k_uniq sorts).
//
// The reference keeps a w-slot ring and rescans it whenever the minimum slides out (two passes over w slots); in SIMT
// some lane rescans at almost every step, so the whole wave would pay ~2w LDS reads per step.  The replay therefore
// uses a monotone deque (hashes non-decreasing front to back, ties kept) that holds exactly the window elements which
// can still become a minimum.  ha_sketch emits an element exactly once iff it equals the minimum of some window that
// ends at or after the first full one -- as the "best" when that is replaced / slides out / the read ends, or as an
// "identical k-mer" copy when a rescan (or the first full window) finds it (sketch.cpp:101-135) -- so here an element
// is emitted the moment it joins the deque's front run.  The one irregular step is the first full window (l == w+k-1):
// copies of the previous partial window's minimum are emitted and that minimum itself is dropped silently if the
// incoming k-mer ties or beats it (sketch.cpp:101-106 run before 116-118 with l < w+k); replicated literally below.
// Dynamic LDS: [w x 64 hashes][read words][w x 64 pos|span][w x 64 time|flag|rev][(k+1) x 64 run lengths].
__host__ __device__ inline size_t sketch_lds_bytes(int w, uint32_t read_words) { return (size_t)read_words * 4 + (size_t)w * 64 * 14 + 64 * 64 + 16; }

struct WordCache { // sequential base access through one cached 16-base word
    const uint32_t *p; uint32_t w; int idx;
    __device__ __forceinline__ uint32_t get(int i) { const int wi = i >> 4; if (wi != idx) { w = p[wi]; idx = wi; } return (w >> ((i & 15) << 1)) & 3u; }
};

__global__ __launch_bounds__(64) void k_sketch(const uint32_t *__restrict__ store, const uint32_t *__restrict__ word_off,
                                               const int32_t *__restrict__ read_len, const uint32_t *__restrict__ mz_off,
                                               fsv_mz *__restrict__ mz, uint32_t *__restrict__ mz_cnt, uint32_t n_reads, int w, int k,
                                               int hpc, uint32_t *__restrict__ warn, const uint8_t *__restrict__ w_per_read, int w_max,
                                               uint32_t lds_words)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    const int lane = threadIdx.x;
    const uint32_t r = blockIdx.x;
    if (r >= n_reads) return;
    const uint32_t woff = word_off[r];
    const int len = read_len[r];
    const uint32_t cap = mz_off[r + 1] - mz_off[r];
    fsv_mz *out = mz + mz_off[r];
    if (w_per_read) w = w_per_read[r];
    uint64_t *d_hash = (uint64_t *)s_dyn;                               // [w_max][64]
    uint32_t *s_words = (uint32_t *)(d_hash + (size_t)w_max * 64);      // [lds_words]
    uint32_t *d_ps = s_words + lds_words;                               // [w_max][64]   pos << 8 | span
    uint16_t *d_tf = (uint16_t *)(d_ps + (size_t)w_max * 64);           // [w_max][64]   (time & 0x3fff) << 2 | rev << 1 | emitted
    uint8_t *q_run = (uint8_t *)(d_tf + (size_t)w_max * 64);            // [64][64]      saturating run lengths
#define D_HASH(j) d_hash[(j) * 64 + lane]
#define D_PS(j) d_ps[(j) * 64 + lane]
#define D_TF(j) d_tf[(j) * 64 + lane]
#define Q_RUN(j) q_run[(j) * 64 + lane]
    const uint32_t nwords = ((uint32_t)len + 15u) >> 4;
    const bool staged = nwords <= lds_words;
    if (staged) for (uint32_t i = lane; i < nwords; i += 64) s_words[i] = store[woff + i];
    __syncthreads();
    WordCache B{staged ? (const uint32_t *)s_words : store + woff, 0u, -1};
    // For even k palindromic k-mers are skipped (sketch.cpp:84), so a fixed warm-up cannot guarantee w entries: lane 0 then
    // replays the whole read from base 0 (exact, 64x less parallel; hifiasm's k = 51 and minimap2's 19 are odd and use k_sketch_fast).
    const bool single = (k & 1) == 0;
    const int c0 = single ? 0 : (int)((long long)len * lane / 64), c1 = single ? (lane == 0 ? len : 0) : (int)((long long)len * (lane + 1) / 64);
    if (c1 <= c0) return;
    int b0 = c0;
    if (hpc) {
        while (b0 > 0 && B.get(b0 - 1) == B.get(b0)) b0--;
        for (int n = 0; n < w + k + 4 && b0 > 0; n++) {
            b0--;
            const uint32_t c = B.get(b0);
            while (b0 > 0 && B.get(b0 - 1) == c) b0--;
        }
    } else {
        b0 = max(0, c0 - (w + k + 4));
    }
    const uint64_t NONE = ~0ull;
    const uint64_t mask = (1ull << k) - 1;
    uint64_t km0 = 0, km1 = 0, km2 = 0, km3 = 0;
    int run_head = 0, run_cnt = 0, span = 0;
    int head = 0, cnt = 0;          // deque = slots (head + j) mod w, j < cnt
    int l = b0 > 0 ? w + k + 1 : 0; // past the start-up phase every "l >= ..." test of the reference holds
    int tail = -1;                  // runs still to replay once the slice is done

#define EMIT_SLOT(sl)                                                                                      \
    do {                                                                                                   \
        const uint32_t ps_ = D_PS(sl); const uint16_t tf_ = D_TF(sl);                                      \
        const int p_ = (int)(ps_ >> 8);                                                                    \
        if (p_ >= c0 && p_ < c1) {                                                                         \
            const uint32_t at_ = atomicAdd(&mz_cnt[r], 1u);                                                \
            if (at_ < cap) { fsv_mz m_; m_.hash = D_HASH(sl); m_.pos = (uint32_t)p_; m_.rev = (uint8_t)((tf_ >> 1) & 1u); m_.span = (uint8_t)(ps_ & 0xffu); m_.pad = 0; out[at_] = m_; } \
            else atomicOr(&warn[r], (uint32_t)FSV_W_MZ_TRUNC);                                             \
        }                                                                                                  \
        D_TF(sl) = (uint16_t)(tf_ | 1u);                                                                   \
    } while (0)
#define WRAP(x) ((x) >= w ? (x) - w : (x))

    int i = b0;
    for (; i < len; i++) {
        if (i >= c1) { if (tail < 0) tail = w + 2; if (tail-- == 0) break; }
        const uint32_t c = B.get(i);
        uint64_t cur_h = NONE; uint32_t cur_ps = 0; uint32_t cur_rev = 0;
        if (hpc) {
            int run = 1;
            while (i + run < len && B.get(i + run) == c) run++;
            i += run - 1;
            const int rs = min(run, 255); // saturating: one run >= 255 puts the span at >= 256 (= no minimizer) either way
            Q_RUN((run_head + run_cnt++) & 63) = (uint8_t)rs;
            span += rs;
            if (run_cnt > k) { span -= Q_RUN(run_head); run_head = (run_head + 1) & 63; run_cnt--; }
        } else {
            span = l + 1 < k ? l + 1 : k;
        }
        km0 = (km0 << 1 | (uint64_t)(c & 1u)) & mask;
        km1 = (km1 << 1 | (uint64_t)(c >> 1)) & mask;
        km2 = km2 >> 1 | (uint64_t)(1u - (c & 1u)) << (k - 1);
        km3 = km3 >> 1 | (uint64_t)(1u - (c >> 1)) << (k - 1);
        if (km1 == km3) continue; // palindrome: not an entry (sketch.cpp:84)
        const int z = km1 < km3 ? 0 : 1;
        ++l;
        if (l >= k && span < 256) {
            cur_h = z ? mix64(km2) + mix64(km3) : mix64(km0) + mix64(km1);
            cur_ps = ((uint32_t)i << 8) | (uint32_t)span;
            cur_rev = (uint32_t)z;
        }
        const int tcur = l & 0x3fff; // entry time, modulo 2^14 (windows are at most 64 entries long)
        // expire what has left the window of the last w entries
        while (cnt > 0 && (((tcur - (int)(D_TF(head) >> 2)) & 0x3fff) >= w)) { head = WRAP(head + 1); cnt--; }
        if (l == w + k - 1 && cnt > 0 && D_HASH(head) != NONE) {
            // first full window (only lanes that replay from base 0 get here): copies of the partial window's minimum
            // are emitted (sketch.cpp:101-106); the minimum itself is lost if the incoming k-mer ties or beats it (:116-118)
            const uint64_t m = D_HASH(head);
            int run = 1;
            while (run < cnt && D_HASH(WRAP(head + run)) == m) run++;
            for (int j = 0; j + 1 < run; j++) { const int sl = WRAP(head + j); EMIT_SLOT(sl); }
            if (cur_h <= m) { const int sl = WRAP(head + run - 1); D_TF(sl) = (uint16_t)(D_TF(sl) | 1u); }
        }
        // keep hashes non-decreasing front to back (ties stay: they are the "identical k-mers")
        while (cnt > 0 && D_HASH(WRAP(head + cnt - 1)) > cur_h) cnt--;
        {
            const int sl = WRAP(head + cnt);
            D_HASH(sl) = cur_h; D_PS(sl) = cur_ps; D_TF(sl) = (uint16_t)((uint32_t)tcur << 2 | cur_rev << 1);
            cnt++;
        }
        if (l >= w + k - 1) {
            const uint64_t m = D_HASH(head);
            if (m != NONE)
                for (int j = 0; j < cnt; j++) {
                    const int sl = WRAP(head + j);
                    if (D_HASH(sl) != m) break;
                    if (!(D_TF(sl) & 1u)) EMIT_SLOT(sl);
                }
        }
    }
    if (i >= len && l < w + k - 1 && cnt > 0 && D_HASH(head) != NONE) {
        // a read shorter than one window: only the last minimum is reported (sketch.cpp:134-135)
        const uint64_t m = D_HASH(head);
        int run = 1;
        while (run < cnt && D_HASH(WRAP(head + run)) == m) run++;
        const int sl = WRAP(head + run - 1);
        EMIT_SLOT(sl);
    }
#undef EMIT_SLOT
#undef WRAP
#undef D_HASH
#undef D_PS
#undef D_TF
#undef Q_RUN
}

// ------------------------------------------------------------------------------------------------ k_uniq
// One workgroup per read: bitonic sort of (hash, pos) in LDS, keep hashes that occur exactly once.
template <int UQ_MAX>
__device__ __forceinline__ void uniq_read(const uint32_t r, fsv_mz *__restrict__ mz, const uint32_t *__restrict__ mz_off, uint32_t *__restrict__ mz_cnt,
                                          uint32_t *__restrict__ warn, const uint32_t *__restrict__ only_changed, uint32_t lo_cnt, uint32_t hi_cnt,
                                          unsigned long long *__restrict__ total, const uint32_t max_occ = 1u)
{
    __shared__ uint64_t s_hash[UQ_MAX];
    __shared__ uint64_t s_pay[UQ_MAX]; // pos | rev << 32 | span << 40
    __shared__ uint32_t s_scan[256];
    const int tid = threadIdx.x;
    if (only_changed && !only_changed[r]) return;   // lists of an unchanged read are already in place
    // The sort holds a read's list in LDS, so the kernel is instantiated for short and for long lists (many reads per CU for
    // the former) and launched once per size class: lo_cnt < raw count <= hi_cnt.  The small class runs first -- it replaces
    // the raw count by the unique count, which can only be smaller, so the large class skips what the small one has done.
    { const uint32_t raw = mz_cnt[r]; if (raw <= lo_cnt || raw > hi_cnt) return; }
    fsv_mz *a = mz + mz_off[r];
    uint32_t n = min(mz_cnt[r], mz_off[r + 1] - mz_off[r]); // k_sketch counts past the cap when it truncates
    const uint32_t n_raw = n;
    if (n > UQ_MAX) { if (tid == 0) atomicOr(&warn[r], (uint32_t)FSV_W_MZ_TRUNC); n = UQ_MAX; }
    uint32_t np = 1;
    while (np < n) np <<= 1;
    for (uint32_t i = tid; i < np; i += 256) {
        if (i < n) { fsv_mz m = a[i]; s_hash[i] = m.hash; s_pay[i] = (uint64_t)m.pos | (uint64_t)m.rev << 32 | (uint64_t)m.span << 40; }
        else { s_hash[i] = ~0ull; s_pay[i] = ~0ull; }
    }
    __syncthreads();
    for (uint32_t sz = 2; sz <= np; sz <<= 1)
        for (uint32_t st = sz >> 1; st > 0; st >>= 1) {
            for (uint32_t t = tid; t < np / 2; t += 256) { // one compare-exchange per thread and trip: pair t = (i, i | st)
                const uint32_t i = ((t & ~(st - 1)) << 1) | (t & (st - 1)), j = i | st;
                const bool up = (i & sz) == 0;
                const uint64_t hi = s_hash[i], hj = s_hash[j], pi = s_pay[i], pj = s_pay[j];
                const bool gt = hi > hj || (hi == hj && (uint32_t)pi > (uint32_t)pj);
                if (gt == up) { s_hash[i] = hj; s_hash[j] = hi; s_pay[i] = pj; s_pay[j] = pi; }
            }
            __syncthreads();
        }
    // unique flags + block compaction (each thread owns a contiguous chunk; its survivors wait in registers until every
    // thread has read its chunk, because they move towards lower indices, i.e. into other threads' chunks)
    const uint32_t per = (n + 255) / 256;
    const uint32_t lo = min(n, tid * per), hi = min(n, lo + per);
    uint64_t kh[UQ_MAX / 256], kp[UQ_MAX / 256];
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t j = 0; j < UQ_MAX / 256; j++) {
        const uint32_t i = lo + j;
        if (i < hi) {
            bool u = (i == 0 || s_hash[i - 1] != s_hash[i]) && (i + 1 >= n || s_hash[i + 1] != s_hash[i]);
            if (max_occ > 1u && !u) {   // the aligner's second seeding of an oversize event keeps hashes that occur up to max_occ times
                uint32_t run = 1;
                for (uint32_t d = 1; d <= max_occ && i >= d && s_hash[i - d] == s_hash[i]; d++) run++;
                for (uint32_t d = 1; d <= max_occ && i + d < n && s_hash[i + d] == s_hash[i]; d++) run++;
                u = run <= max_occ;
            }
            if (u) { kh[cnt] = s_hash[i]; kp[cnt] = s_pay[i]; cnt++; }
        }
    }
    s_scan[tid] = cnt;
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int i = 0; i < 256; i++) { uint32_t c = s_scan[i]; s_scan[i] = acc; acc += c; }
        mz_cnt[r] = acc;
        if (total) {   // statistics of the launch: unique minimizers; minimizers the sketch produced; bases of the reads it sketched
            atomicAdd(total, (unsigned long long)acc);
            atomicAdd(total + 3, (unsigned long long)n_raw);                                     // CT_MZRAW (asm.hip)
            atomicAdd(total + 4, (unsigned long long)(mz_off[r + 1] - mz_off[r] - 64u));         // CT_BASES: a slot holds len + 64 entries
        }
    }
    __syncthreads();
    const uint32_t m = mz_cnt[r];
    const uint32_t o0 = s_scan[tid];
    const uint32_t slot_cap = mz_off[r + 1] - mz_off[r];
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < UQ_MAX / 256; j++)
        if (j < cnt) { s_hash[o0 + j] = kh[j]; s_pay[o0 + j] = kp[j]; }
    __syncthreads();
    // [0, m): sorted by hash (the "target" role: binary-searched)
    for (uint32_t i = tid; i < m; i += 256) { fsv_mz x; x.hash = s_hash[i]; const uint64_t p = s_pay[i]; x.pos = (uint32_t)p; x.rev = (uint8_t)(p >> 32); x.span = (uint8_t)(p >> 40); x.pad = 0; a[i] = x; }
    // [m, 2m): the same minimizers sorted by position (the "query" role: anchors then come out in query order and the
    // chain kernels need no per-pair sort)
    if (2 * m <= slot_cap) {
        uint32_t mp = 1;
        while (mp < m) mp <<= 1;
        for (uint32_t i = m + tid; i < mp; i += 256) { s_hash[i] = ~0ull; s_pay[i] = ~0ull; }
        __syncthreads();
        for (uint32_t sz = 2; sz <= mp; sz <<= 1)
            for (uint32_t st = sz >> 1; st > 0; st >>= 1) {
                for (uint32_t t = tid; t < mp / 2; t += 256) {
                    const uint32_t i = ((t & ~(st - 1)) << 1) | (t & (st - 1)), j = i | st;
                    const bool up = (i & sz) == 0;
                    const uint64_t pi = s_pay[i], pj = s_pay[j];
                    bool gt = (uint32_t)pi > (uint32_t)pj || ((uint32_t)pi == (uint32_t)pj && pi > pj);
                    if (pi == ~0ull && pj != ~0ull) gt = true; else if (pj == ~0ull) gt = false;
                    if (gt == up) { const uint64_t hi2 = s_hash[i], hj = s_hash[j]; s_hash[i] = hj; s_hash[j] = hi2; s_pay[i] = pj; s_pay[j] = pi; }
                }
                __syncthreads();
            }
        for (uint32_t i = tid; i < m; i += 256) { fsv_mz x; x.hash = s_hash[i]; const uint64_t p = s_pay[i]; x.pos = (uint32_t)p; x.rev = (uint8_t)(p >> 32); x.span = (uint8_t)(p >> 40); x.pad = 0; a[m + i] = x; }
    } else if (tid == 0) atomicOr(&warn[r], (uint32_t)FSV_W_INTERNAL); // cannot happen: at most one minimizer per base and slots hold len + 64
}

template <int UQ_MAX>
__global__ __launch_bounds__(256) void k_uniq(fsv_mz *__restrict__ mz, const uint32_t *__restrict__ mz_off, uint32_t *__restrict__ mz_cnt,
                                              uint32_t *__restrict__ warn, const uint32_t *__restrict__ only_changed = nullptr,
                                              uint32_t lo_cnt = 0u, uint32_t hi_cnt = 0xffffffffu, unsigned long long *__restrict__ total = nullptr,
                                              uint32_t max_occ = 1u)
{
    uniq_read<UQ_MAX>(blockIdx.x, mz, mz_off, mz_cnt, warn, only_changed, lo_cnt, hi_cnt, total, max_occ);
}

// the same for a size class that is usually empty (lists above 1 024 entries in a HiFi batch): a few blocks walk all reads, so the
// 64 KB of LDS a block of this instantiation needs are claimed a few hundred times, not once per read (under three lanes the
// one-block-per-read launch averaged 2.8 ms against 0.01 alone: every block waited for LDS only to find its read in the other class)
template <int UQ_MAX>
__global__ __launch_bounds__(256) void k_uniq_walk(fsv_mz *__restrict__ mz, const uint32_t *__restrict__ mz_off, uint32_t *__restrict__ mz_cnt,
                                                   uint32_t *__restrict__ warn, const uint32_t *__restrict__ only_changed, uint32_t lo_cnt, uint32_t hi_cnt,
                                                   unsigned long long *__restrict__ total, uint32_t n_reads)
{
    for (uint32_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
        uniq_read<UQ_MAX>(r, mz, mz_off, mz_cnt, warn, only_changed, lo_cnt, hi_cnt, total);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ k_chain
// wave-wide max, uniform result.  __shfl_xor goes through the LDS crossbar (ds_bpermute, ~6 dependent round trips);
// DPP row operations reduce each 16-lane row in four VALU ops and the four row results are read as scalars.
__device__ __forceinline__ int wave_max_i32(int v)
{
    const int lowest = -2147483647 - 1;
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x141, 0xF, 0xF, false)); // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x140, 0xF, 0xF, false)); // row_mirror
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ long long wave_max_i64(long long v)
{
    for (int off = 32; off > 0; off >>= 1) { long long o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

__device__ __forceinline__ int thr_for_len(int x_len, const uint8_t *__restrict__ thr_tab) { return thr_tab[x_len]; }

struct ChainArgs {
    const uint32_t *store;
    const uint32_t *word_off;
    const int32_t *read_len;
    const uint32_t *set_start;   // n_sets + 1
    const uint32_t *pair_base;   // n_sets + 1: ordered-pair slots
    const uint32_t *upair_base;  // n_sets + 1: unordered pairs (one block each)
    const uint32_t *pair_list;   // optional: block b works on unordered pair pair_list[b] (nullptr: pair b)
    const uint32_t *n_list_dev;  // with pair_list: its length, left on the device by the kernel that built it (blocks beyond it return)
    const uint4 *upair_tab;      // per unordered pair: {first read of the set, q | t << 16, slot (q,t), slot (t,q)}  (k_pair_tab)
    int32_t amax;                // anchors per pair held in LDS (multiple of 64, <= FSV_AMAX): sizes the dynamic LDS
    const fsv_mz *mz;
    const uint32_t *mz_off;
    const uint32_t *mz_cnt;
    fsv_ovl *ovl;                // one slot per ordered pair
    fsv_wtask *tasks;
    uint32_t *task_counter;
    uint32_t task_cap;
    uint32_t *overflow;          // set to 1 when the task array is full
    uint32_t *warn;              // per read
    uint32_t *set_cols;          // per read, used at the first read of every set: K5 columns emitted for the set (statistics)
    const uint8_t *thr_tab;      // 376 entries: threshold for a window of that length
    uint32_t n_sets;
    int32_t k_score, min_anchors, min_ovlp, bw, emit_tasks;
    int32_t primary_only;        // 1 (without tasks only): the slot of (q, t) alone is written -- the overlap of t on q is chained from t's side by a second launch
    unsigned long long *stamps;  // diagnostic (FSV_CHAIN_STAMPS=1): shader cycles per phase summed over the waves, else nullptr
    uint32_t *wide_list, *n_wide; // pairs whose two lists both have more than amax entries (only they can have more than amax anchors) are
                                  // set aside here and chained by k_chain_wide_list with the large tile; nullptr: chain every pair here
};

// The unordered pairs of every set, enumerated once per batch: block b of k_chain reads one 16-byte record instead of
// searching the set table and inverting the triangular index (a dozen dependent global loads per block).
// the pair table with the roles of the two reads swapped (the final pass's gapped re-chain chains a pair from either side)
__global__ void k_pair_tab_swap(const uint4 *__restrict__ tab, uint32_t n_upairs, uint4 *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_upairs) return;
    const uint4 v = tab[i];
    out[i] = make_uint4(v.x, (v.y >> 16) | (v.y << 16), v.w, v.z);
}

__global__ void k_pair_tab(const uint32_t *__restrict__ set_start, const uint32_t *__restrict__ pair_base, const uint32_t *__restrict__ upair_base,
                           uint32_t n_sets, uint32_t n_upairs, uint4 *__restrict__ tab, uint32_t *__restrict__ pair_read)
{
    const uint32_t up = blockIdx.x * blockDim.x + threadIdx.x;
    if (up >= n_upairs) return;
    uint32_t lo = 0, hi = n_sets;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (upair_base[mid] <= up) lo = mid; else hi = mid; }
    const uint32_t s = lo, r0 = set_start[s], ns = set_start[s + 1] - r0;
    const uint32_t idx = up - upair_base[s];
    // row q holds the pairs (q, q+1..ns-1): rows start at q*(2ns-q-1)/2
    uint32_t q = (uint32_t)((2.0 * ns - 1.0 - sqrt((2.0 * ns - 1.0) * (2.0 * ns - 1.0) - 8.0 * (double)idx)) * 0.5);
    while (q > 0 && (uint64_t)q * (2ull * ns - q - 1) / 2 > idx) q--;
    while ((uint64_t)(q + 1) * (2ull * ns - q - 2) / 2 <= idx) q++;
    const uint32_t t = q + 1 + (idx - (uint32_t)((uint64_t)q * (2ull * ns - q - 1) / 2));
    tab[up] = make_uint4(r0, q | t << 16, pair_base[s] + q * (ns - 1) + (t - 1), pair_base[s] + t * (ns - 1) + q);
    // the query read of either ordered slot (k_bnd_tasks: one load instead of a search of the set table)
    pair_read[pair_base[s] + q * (ns - 1) + (t - 1)] = r0 + q;
    pair_read[pair_base[s] + t * (ns - 1) + q] = r0 + t;
}

// One wavefront per UNORDERED read pair (q < t) of a set: the chain is computed with q as the query and the overlap of t on
// q is its mirror image (oracle/asm.c collect_overlaps); both ordered slots and both window-task lists are written here.
//
// LDS per anchor.  SHORT (every read of the batch shorter than 65 536 bases -- all HiFi data): 12 B -- the anchor's two
// positions as 16-bit halves of one word, score / run start / predecessor / chain entry as 16-bit values (a chain of at most
// 1 024 anchors scores at most 1 024 x 63), the anchors' strands in a 128-byte bitmap, and the target's sorted hashes staged
// in the 8 B the DP arrays do not need yet (position / span / strand of a hit come from the L2-resident list).  Round 1 used
// 24 B (64-bit keys, 32-bit DP arrays, 12-byte staged records), which capped the kernel at 1-2 waves per SIMD on the batch's
// longest lists; the long layout is kept for batches with a read of 65 536 bases or more.
#define FSV_CHAIN_QR 12   // the query list sits in registers when it has at most 64 x this many entries (768: reads up to ~27 kb; 16 would cost the third wave per SIMD)
template <bool SHORT>
__device__ __forceinline__ void chain_pair(const ChainArgs &A, unsigned char *s_raw, const uint4 pt, const int lenq, const int lent, const int nq, const int nt,
                                           const fsv_mz *mq, const fsv_mz *mt, const uint4 (&qa)[FSV_CHAIN_QR])
{
    const int AMAX = A.amax;
    using key_t = typename std::conditional<SHORT, uint32_t, uint64_t>::type;   // qe << 16 | te   or   qe << 32 | te
    using dp_t = typename std::conditional<SHORT, uint16_t, int32_t>::type;
    constexpr int KSH = SHORT ? 16 : 32;
    constexpr uint64_t KMASK = SHORT ? 0xffffull : 0xffffffffull;
    key_t *const s_key = (key_t *)s_raw;
    unsigned char *const s_rest = s_raw + sizeof(key_t) * (size_t)AMAX;
    // SHORT: s_f | s_ind | s_aux | s_chain, 2 B each (8 B: the staged target hashes lie over all four)
    // long:  s_f | s_ind | (4 B only used by the staged records) | s_aux | s_chain
    dp_t *const s_f = (dp_t *)s_rest, *const s_ind = s_f + AMAX;
    uint16_t *const s_aux = (uint16_t *)(s_rest + (SHORT ? 4 : 12) * (size_t)AMAX);   // long: t span | strand << 8; then the predecessor index
    uint16_t *const s_chain = s_aux + AMAX;
    uint32_t *const s_strand = (uint32_t *)(s_rest + (SHORT ? 8 : 16) * (size_t)AMAX);  // SHORT only: one strand bit per anchor (AMAX / 8 B)
#define KEY_Q(i) ((int)((uint64_t)s_key[i] >> KSH))
#define KEY_T(i) ((int)((uint64_t)s_key[i] & KMASK))
#define MAKE_KEY(q_, t_) ((key_t)(((uint64_t)(uint32_t)(q_) << KSH) | (uint64_t)(uint32_t)(t_)))
    int lane_ = threadIdx.x;
    // (opaque to the optimiser: called in a loop, the compiler otherwise hoists every lane-derived constant of the body out of it
    // and holds them in ~80 extra registers -- two waves per SIMD instead of three)
    asm volatile("" : "+v"(lane_));
    const int lane = lane_;
    const uint32_t q = pt.y & 0xffffu, t = pt.y >> 16;
    const uint32_t p = pt.z, pm = pt.w;     // ordered slots (q, t) and (t, q)
    const uint32_t rq = pt.x + q, rt = pt.x + t;
    fsv_ovl o;
    o.q = q; o.t = t; o.x_s = o.x_e = o.y_s = o.y_e = 0; o.score = 0; o.n_chain = 0; o.chain_off = 0; o.first_win = 0; o.n_win = 0;
    o.align_len = 0; o.err_sum = 0; o.rev = 0; o.is_match = 0; o.exact = 0; o.valid = 0;
    fsv_ovl om = o; // the mirrored overlap (t on q)
    om.q = t; om.t = q;
#define PUT_BOTH() do { if (lane == 0) { A.ovl[p] = o; if (!A.primary_only) A.ovl[pm] = om; } } while (0)
    const bool stamped = A.stamps && (blockIdx.x & 63u) == 0u;   // one block in 64: the atomics must not become the load
    unsigned long long tm = stamped ? __builtin_amdgcn_s_memtime() : 0ull;
#define CH_MARK(i_) do { if (stamped) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) { atomicAdd(&A.stamps[i_], t_ - tm); atomicAdd(&A.stamps[8 + (i_)], 1ull); } tm = t_; } } while (0)

    // 1. anchors: every q minimizer is looked up in t's sorted unique list.  All global loads are issued up front -- t's
    //    hashes go to LDS (the long layout also stages {pos, span, strand}: 12 B per entry in the DP arrays, free until the
    //    DP), q's records to registers (16 B per lane per 64 minimizers) -- so a pair pays one memory latency instead of two
    //    per batch of 64 lookups; the ~10 probes of a lookup are LDS reads.
    uint64_t *s_th = (uint64_t *)s_rest;
    uint32_t *s_tp = (uint32_t *)(s_rest + 8 * (size_t)AMAX);   // long layout only (staging them for SHORT too -- 16 B per anchor, 9 pairs per CU -- was slower)
    const bool t_in_lds = nt <= AMAX && lent < (1 << 23);
    const uint4 *mq4 = (const uint4 *)mq, *mt4 = (const uint4 *)mt;
    constexpr int QR = FSV_CHAIN_QR;
    const bool q_in_regs = nq <= 64 * FSV_CHAIN_QR;   // (the caller loaded them)
    if (t_in_lds) {
        // eight loads in flight per lane: written as a plain loop the compiler waits for every load before it issues the next
        // (s_waitcnt vmcnt(0) in front of each LDS write) -- seven dependent round trips for a 15 kb read's list, which was
        // most of the kernel's time
        for (int base = 0; base < nt; base += 512) {
            if (SHORT) {
                uint2 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { const int i = base + u * 64 + lane; if (i < nt) v[u] = *reinterpret_cast<const uint2 *>(mt4 + i); }
#pragma unroll
                for (int u = 0; u < 8; u++) { const int i = base + u * 64 + lane; if (i < nt) s_th[i] = (uint64_t)v[u].x | (uint64_t)v[u].y << 32; }
            } else {
                uint4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { const int i = base + u * 64 + lane; if (i < nt) v[u] = mt4[i]; }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = base + u * 64 + lane;
                    if (i < nt) {
                        s_th[i] = (uint64_t)v[u].x | (uint64_t)v[u].y << 32;
                        s_tp[i] = v[u].z | (v[u].w & 0xffu) << 31 | ((v[u].w >> 8) & 0xffu) << 23; // pos < 2^23 | span << 23 | strand << 31
                    }
                }
            }
        }
    }
    if (SHORT) for (int i = lane; i < AMAX / 32; i += 64) s_strand[i] = 0u;
    __syncthreads();
    // the hashes are uniform, so their top six bits cut the sorted list into 64 buckets of a few entries each: one search per
    // lane finds the bucket bounds, and a lookup then needs ~3 probes instead of log2(nt) ~ 9 (the lookups were nearly all of
    // the kernel's instructions: every q minimizer against every t list of the set, overlapping or not)
    __shared__ uint32_t s_bk[65];
    if (t_in_lds) {
        int l2 = 0, h2 = nt;
        while (l2 < h2) { const int mid = (l2 + h2) >> 1; if ((uint32_t)(s_th[mid] >> 58) < (uint32_t)lane) l2 = mid + 1; else h2 = mid; }
        s_bk[lane] = (uint32_t)l2;
        if (lane == 0) s_bk[64] = (uint32_t)nt;
        __syncthreads();
    }
    CH_MARK(0);
    int n = 0, nrev = 0, nfwd = 0;
    auto commit = [&](bool hit, key_t key, uint32_t srev, uint32_t tspan) {
        uint64_t m = __ballot(hit);
        int at = n + __popcll(m & ((1ull << lane) - 1));
        if (hit && at < AMAX) {
            s_key[at] = key;
            if (SHORT) { if (srev) atomicOr(&s_strand[at >> 5], 1u << (at & 31)); }
            else s_aux[at] = (uint16_t)(tspan | (srev << 8));
        }
        nrev += __popcll(__ballot(hit && srev));
        nfwd += __popcll(__ballot(hit && !srev));
        n += __popcll(m);
    };
    auto lookup = [&](int i, const uint4 av) {
        bool hit = false; key_t key = 0; uint32_t srev = 0, tspan = 0;
        if (i < nq) {
            const uint64_t ah = (uint64_t)av.x | (uint64_t)av.y << 32;
            const uint32_t arev = av.w & 0xffu;
            // a reverse-strand anchor is kept as hifiasm chains such a pair (Hash_Table.cpp:619-676, x_pos_strand = 1): the query on its
            // reverse strand -- the k-mer's last base there -- and the target forward; the chain's indel budget runs from that end
            const uint32_t qrev = (uint32_t)(lenq - 1) - (av.z - ((av.w >> 8) & 0xffu) + 1);
            int l2 = 0, h2 = nt;
            if (t_in_lds) {
                l2 = (int)s_bk[av.y >> 26]; h2 = (int)s_bk[(av.y >> 26) + 1];
                while (l2 < h2) { int mid = (l2 + h2) >> 1; if (s_th[mid] < ah) l2 = mid + 1; else h2 = mid; }
                if (l2 < nt && s_th[l2] == ah) {
                    uint32_t tpos;
                    if (SHORT) { const uint4 b = mt4[l2]; tpos = b.z; srev = arev ^ (b.w & 0xffu); tspan = (b.w >> 8) & 0xffu; }
                    else { const uint32_t tp = s_tp[l2]; tpos = tp & 0x7fffffu; srev = arev ^ (tp >> 31); tspan = (tp >> 23) & 0xffu; }
                    hit = true; key = MAKE_KEY(srev ? qrev : av.z, tpos);
                }
            } else {
                while (l2 < h2) { int mid = (l2 + h2) >> 1; if (mt[mid].hash < ah) l2 = mid + 1; else h2 = mid; }
                if (l2 < nt && mt[l2].hash == ah) {
                    fsv_mz b = mt[l2];
                    srev = arev ^ b.rev; tspan = b.span;
                    hit = true; key = MAKE_KEY(srev ? qrev : av.z, b.pos);
                }
            }
        }
        commit(hit, key, srev, tspan);
    };
    if (q_in_regs && t_in_lds) {
        // four batches of 64 lookups at a time: their probe chains are independent, so the four LDS reads of a step (and the
        // four fetches of the hits' records) are in flight together -- one batch at a time a wave spent ~1 700 cycles per batch
        // on dependent LDS / memory latency (FSV_CHAIN_STAMPS)
#pragma unroll
        for (int u0 = 0; u0 < QR; u0 += 4) {
            if (u0 * 64 < nq) {
                int l[4], h[4];
                uint64_t ah[4];
                bool val[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint4 av = qa[u0 + j];
                    val[j] = (u0 + j) * 64 + lane < nq;
                    ah[j] = (uint64_t)av.x | (uint64_t)av.y << 32;
                    const uint32_t b = av.y >> 26;
                    l[j] = val[j] ? (int)s_bk[b] : 0; h[j] = val[j] ? (int)s_bk[b + 1] : 0;
                }
                for (;;) {
                    bool more = false;
                    uint64_t v[4];
                    int mid[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) { mid[j] = (l[j] + h[j]) >> 1; v[j] = s_th[mid[j]]; }   // (mid <= nt <= AMAX: inside the tile)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const bool act = l[j] < h[j], lt = v[j] < ah[j];
                        l[j] = act && lt ? mid[j] + 1 : l[j]; h[j] = act && !lt ? mid[j] : h[j];
                        more |= act;
                    }
                    if (!__any(more)) break;
                }
                uint64_t c[4];
#pragma unroll
                for (int j = 0; j < 4; j++) c[j] = s_th[l[j]];
                bool hit[4];
                uint32_t tz[4] = {0, 0, 0, 0}, tw[4] = {0, 0, 0, 0};   // position; strand | span << 8
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    hit[j] = val[j] && l[j] < nt && c[j] == ah[j];
                    if (SHORT) { if (hit[j]) { const uint2 b = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(mt4 + l[j]) + 8); tz[j] = b.x; tw[j] = b.y; } }
                    else if (hit[j]) { const uint32_t tp = s_tp[l[j]]; tz[j] = tp & 0x7fffffu; tw[j] = (tp >> 31) | ((tp >> 23) & 0xffu) << 8; }
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if ((u0 + j) * 64 < nq) {
                        const uint4 av = qa[u0 + j];
                        const uint32_t qrev = (uint32_t)(lenq - 1) - (av.z - ((av.w >> 8) & 0xffu) + 1);
                        const uint32_t srev = (av.w & 0xffu) ^ (tw[j] & 0xffu);
                        commit(hit[j], MAKE_KEY(srev ? qrev : av.z, tz[j]), hit[j] ? srev : 0u, (tw[j] >> 8) & 0xffu);
                    }
            }
        }
    } else
        for (int base = 0; base < nq; base += 64) { const int i = base + lane; lookup(i, i < nq ? mq4[i] : make_uint4(0, 0, 0, 0)); }
    if (n > AMAX) { if (lane == 0) atomicOr(&A.warn[rq], (uint32_t)FSV_W_ANCHOR_TRUNC); n = AMAX; }
    __syncthreads();
    CH_MARK(1);
    // 2. majority strand, compaction
    const int rev = nrev > nfwd;
    int m2 = 0;
    for (int base = 0; base < n; base += 64) {
        int i = base + lane;
        bool keep = false; key_t key = 0;
        if (i < n) {
            key = s_key[i];
            keep = (SHORT ? (int)((s_strand[i >> 5] >> (i & 31)) & 1u) : (int)(s_aux[i] >> 8)) == rev;
        }
        uint64_t m = __ballot(keep);
        int at = m2 + __popcll(m & ((1ull << lane) - 1));
        __syncthreads();
        if (keep) s_key[at] = key;
        m2 += __popcll(m);
        __syncthreads();
    }
    n = m2;
    if (n < A.min_anchors) { PUT_BOTH(); CH_MARK(2); return; }
    // 3. anchors are in query order: q's minimizers were walked by position and both compactions keep the order (query positions
    //    are distinct, so (qe, te) order == qe order) -- for a reverse-strand pair that is decreasing order on the query's reverse
    //    strand, so the list is turned around
    if (rev) {
        for (int i = lane; i < n / 2; i += 64) { const key_t a0 = s_key[i], a1 = s_key[n - 1 - i]; s_key[i] = a1; s_key[n - 1 - i] = a0; }
        __syncthreads();
    }
    // 4. chain DP: lane l examines predecessor i-1-l (nearest first on ties).
    //    Fast path: when every anchor sits on one diagonal (error-free reads: correction rounds 2, 3 and the final pass)
    //    the DP provably links each anchor to its nearest predecessor -- gap 0 means no indel penalty, and
    //    f[i-1] + min(d_i,k) >= f[j] + min(qe_i - qe_j, k) for every j < i-1 because min(.,k) is sub-additive, with the
    //    nearest predecessor winning ties -- so the chain is the whole list and the score a running sum.
    CH_MARK(3);
    bool colinear;
    {
        const int d0 = KEY_T(0) - KEY_Q(0);
        bool same = true;
        for (int i = lane; i < n; i += 64) same = same && (KEY_T(i) - KEY_Q(i) == d0);
        colinear = __all(same);
    }
    if (colinear) {
        int acc = 0;
        for (int i = 1 + lane; i < n; i += 64) acc += min(KEY_Q(i) - KEY_Q(i - 1), A.k_score);
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        for (int i = lane; i < n; i += 64) { s_aux[i] = (uint16_t)(i == 0 ? 0xffff : i - 1); s_f[i] = (dp_t)(i == n - 1 ? A.k_score + acc : 0); }
        __syncthreads();
    } else {
        // Reads with errors: the anchors leave the diagonal at every indel, and the DP of the reference (Hash_Table.cpp:425-616 as
        // restated in oracle/asm.c: look back 64 anchors, link to the best-scoring predecessor, the nearer one on ties) is a chain
        // of n dependent steps.  But it almost always links an anchor to the one just before it, so 64 anchors are settled at once:
        //   hypothesis   every anchor of the block links to its predecessor; the chain's indel sum, span and score are then
        //                prefix sums over the block (three wave scans);
        //   proof        cand(i, j) <= f[j] + k for any other predecessor j, so only the j with f[j] + k > f[i] can beat the
        //                hypothesis (usually none, or i-2): those candidates are evaluated exactly as the DP does;
        //   repair       the first anchor whose hypothesis fails (an illegal link, a better candidate) and the few behind it go
        //                through the sequential step -- the 64 predecessors in registers, lane l = anchor i-1-l, handed on by
        //                DPP wave_shr -- and the blocks resume after them.
        // By induction over the anchors the result is the sequential DP's, bit for bit (tests/test_gpu_asm.py against the oracle).
        dp_t *const s_sl = SHORT ? (dp_t *)s_chain : (dp_t *)(s_rest + 8 * (size_t)AMAX);   // chain span per anchor (free arrays during the DP)
        const int kk = A.k_score;
        // candidate (i <- j): score or -1, with the chain's indel sum / span it would give
        auto eval = [&](int qe, int te, int qj, int tj, int indj, int slj, int fj, int &ti, int &tl) -> int {
            const int dq = qe - qj, dt = te - tj;
            if (dq <= 0 || dt <= 0) return -1;
            const int gap = dq > dt ? dq - dt : dt - dq;
            ti = indj + gap; tl = slj + dq;
            // 64-bit divisions cost ~150 VALU ops on gfx950; the operands fit 32 bits for every read below 2^17 bases
            // (ti <= tl*bw/1000 <= 2621, sc <= 63): same quotient either way
            if (tl < (1 << 17) && gap < (1 << 17) && A.bw <= 20) {
                if ((uint32_t)ti * 1000u > (uint32_t)tl * (uint32_t)A.bw) return -1;
                int sc = min(min(dq, dt), kk);
                if (ti) sc -= (int)(((uint32_t)ti * (uint32_t)sc * 1000u) / ((uint32_t)tl * (uint32_t)A.bw));
                return sc + fj;
            }
            if ((long long)ti * 1000 > (long long)tl * A.bw) return -1;
            int sc = min(min(dq, dt), kk);
            if (ti) sc -= (int)(((long long)ti * sc * 1000) / ((long long)tl * A.bw));
            return sc + fj;
        };
        auto scan_add = [&](int v) { for (int off = 1; off < 64; off <<= 1) { const int o2 = __shfl_up(v, off, 64); if (lane >= off) v += o2; } return v; };
        if (lane == 0) { s_f[0] = (dp_t)kk; s_aux[0] = 0xffff; s_ind[0] = 0; s_sl[0] = 0; }
        __syncthreads();
        int i0 = 1;
        while (i0 < n) {
            const int nb = min(64, n - i0), i = i0 + lane;
            const bool in = lane < nb;
            int qe = 0, te = 0, dq = 0, dt = 0, gap = 0;
            if (in) { qe = KEY_Q(i); te = KEY_T(i); dq = qe - KEY_Q(i - 1); dt = te - KEY_T(i - 1); gap = dq > dt ? dq - dt : dt - dq; }
            const int ti = (int)s_ind[i0 - 1] + scan_add(gap), tl = (int)s_sl[i0 - 1] + scan_add(dq);
            bool legal = in && dq > 0 && dt > 0;
            int sc = 0;
            if (legal) {
                int t2, l2;
                const int c = eval(qe, te, qe - dq, te - dt, ti - gap, tl - dq, 0, t2, l2);
                legal = c >= 0; sc = c;
            }
            const int fi = (int)s_f[i0 - 1] + scan_add(legal ? sc : 0);
            bool bad = in && !(legal && fi > kk);
            __syncthreads();
            if (in) { s_f[i] = (dp_t)fi; s_ind[i] = (dp_t)ti; s_sl[i] = (dp_t)tl; }
            __syncthreads();
            // The look-back stops where no earlier anchor can matter any more: P[j] = max f over the 64 anchors before the block
            // and the block up to j never decreases with j, so once P[i-d] + k <= f[i] nothing at distance d or beyond can beat
            // the hypothesis.  Scores grow by ~35 an anchor, so that is after two or three steps -- the loop used to run all 63
            // (a dependent LDS read each) whenever the block had that many predecessors: most of the round-1 DP's time.
            // f and P of the 128 anchors sit in registers; distance d is a lane rotation.
            const int pj0 = i0 - 64 + lane;
            const int pf = pj0 >= 0 ? (int)s_f[pj0] : 0, cf = in ? fi : 0;
            int pP = pf, cP = cf;
            for (int off = 1; off < 64; off <<= 1) {
                const int o1 = __shfl_up(pP, off, 64), o2 = __shfl_up(cP, off, 64);
                if (lane >= off) { pP = max(pP, o1); cP = max(cP, o2); }
            }
            cP = max(cP, __shfl(pP, 63, 64));
            bool live = in && !bad;
            for (int d = 2; d <= 64; d++) {
                const int j = i - d, src = (lane - d) & 63;
                const int f_c = __shfl(cf, src, 64), f_p = __shfl(pf, src, 64), P_c = __shfl(cP, src, 64), P_p = __shfl(pP, src, 64);
                const int fj = lane >= d ? f_c : f_p, Pj = lane >= d ? P_c : P_p;
                live = live && !bad && j >= 0 && Pj + kk > fi;
                if (!__any(live)) break;
                const bool need = live && fj + kk > fi;
                if (__any(need)) {
                    if (need) {
                        int t2, l2;
                        if (eval(qe, te, KEY_Q(j), KEY_T(j), (int)s_ind[j], (int)s_sl[j], fj, t2, l2) > fi) bad = true;
                    }
                }
            }
            const uint64_t badm = __ballot(bad);
            const int good = badm ? (int)__ffsll((long long)badm) - 1 : nb;     // anchors i0 .. i0 + good - 1 stand
            if (lane < good) s_aux[i] = (uint16_t)(i - 1);
            __syncthreads();
            i0 += good;
            if (good == nb) continue;
            // sequential steps for the anchor that broke the hypothesis and up to seven behind it
            const int s1 = min(n, i0 + 8);
            int rq = 0, rt = 0, rind = 0, rsl = 0, rf = 0;
            { const int j = i0 - 1 - lane; if (j >= 0) { rq = KEY_Q(j); rt = KEY_T(j); rind = (int)s_ind[j]; rsl = (int)s_sl[j]; rf = (int)s_f[j]; } }
            for (int is = i0; is < s1; is++) {
                const int qe2 = KEY_Q(is), te2 = KEY_T(is);
                const int j = is - 1 - lane;
                int cand = -1, ti2 = 0, tl2 = 0;
                if (j >= 0) cand = eval(qe2, te2, rq, rt, rind, rsl, rf, ti2, tl2);
                // pack so that the max prefers the higher score, then the nearer predecessor
                const int packed = cand < 0 ? -1 : cand * 64 + (63 - lane);
                const int bestp = wave_max_i32(packed);
                const int bests = bestp < 0 ? -1 : bestp >> 6;
                int nf = kk, nind = 0, nsl = 0, npred = 0xffff;
                if (bests > kk) {
                    const int wl = 63 - (bestp & 63);
                    nf = bests; npred = is - 1 - wl;
                    nind = __builtin_amdgcn_readlane(ti2, wl); nsl = __builtin_amdgcn_readlane(tl2, wl);
                }
                if (lane == 0) { s_f[is] = (dp_t)nf; s_aux[is] = (uint16_t)npred; s_ind[is] = (dp_t)nind; s_sl[is] = (dp_t)nsl; }
                rq = __builtin_amdgcn_update_dpp(qe2, rq, 0x138, 0xF, 0xF, false);    // wave_shr:1, lane 0 <- the new anchor
                rt = __builtin_amdgcn_update_dpp(te2, rt, 0x138, 0xF, 0xF, false);
                rind = __builtin_amdgcn_update_dpp(nind, rind, 0x138, 0xF, 0xF, false);
                rsl = __builtin_amdgcn_update_dpp(nsl, rsl, 0x138, 0xF, 0xF, false);
                rf = __builtin_amdgcn_update_dpp(nf, rf, 0x138, 0xF, 0xF, false);
            }
            __syncthreads();
            i0 = s1;
        }
        __syncthreads();
    }
    CH_MARK(4);
    // 5. best chain end: highest score, smallest index on ties
    long long bk = -1;
    for (int i = lane; i < n; i += 64) { long long v = (long long)s_f[i] * 4096 + (4095 - i); bk = v > bk ? v : bk; }
    bk = wave_max_i64(bk);
    const int best = 4095 - (int)(bk & 4095);
    // 6. walk back, chain stored end-to-start in s_chain.  A step-by-step walk is ~n dependent LDS reads; instead every
    //    anchor learns the start of its run of "predecessor == previous anchor" links (a max-scan; s_ind is free after the
    //    DP) and the walk copies whole runs, one dependent step per break in the chain.
    int cnt = 0;
    if (colinear) {
        for (int e = lane; e <= best; e += 64) s_chain[e] = (uint16_t)(best - e);
        cnt = best + 1;
    } else {
        int carry = 0;
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            int v = (i < n && i > 0 && s_aux[i] == (uint16_t)(i - 1)) ? -1 : i; // run start candidate
            if (i >= n) v = -1;
            for (int off = 1; off < 64; off <<= 1) { const int o2 = __shfl_up(v, off, 64); if (lane >= off) v = max(v, o2); }
            v = max(v, carry);
            if (i < n) s_ind[i] = (dp_t)v;
            carry = __shfl(v, 63, 64);
        }
        __syncthreads();
        int c = best;
        while (c != 0xffff) {
            const int r = s_ind[c];
            for (int e = lane; e <= c - r; e += 64) s_chain[cnt + e] = (uint16_t)(c - e);
            cnt += c - r + 1;
            c = s_aux[r];
        }
    }
    __syncthreads();
    CH_MARK(5);
    if (cnt < A.min_anchors) { PUT_BOTH(); return; }
    const int first = s_chain[cnt - 1];
    int xs = KEY_Q(first), ys = KEY_T(first);
    int xe = KEY_Q(best), ye = KEY_T(best);
    { int m = min(xs, ys); xs -= m; ys -= m; int r = min(lenq - 1 - xe, lent - 1 - ye); xe += r; ye += r; }
    if (xe - xs + 1 < A.min_ovlp) { PUT_BOTH(); return; }
    const int score_best = s_f[best];
    if (rev) {
        // everything downstream works with the query forward and the target on its reverse strand: mirror the overlap and every
        // anchor, and turn the chain list around so that it still runs end-to-start in query order
        { const int t0 = xs; xs = (lenq - 1) - xe; xe = (lenq - 1) - t0; }
        { const int t0 = ys; ys = (lent - 1) - ye; ye = (lent - 1) - t0; }
        __syncthreads();
        for (int i = lane; i < n; i += 64) {
            const int kq = KEY_Q(i), kt2 = KEY_T(i);
            s_key[i] = MAKE_KEY((lenq - 1) - kq, (lent - 1) - kt2);
        }
        for (int e = lane; e < cnt / 2; e += 64) { const uint16_t c0 = s_chain[e], c1 = s_chain[cnt - 1 - e]; s_chain[e] = c1; s_chain[cnt - 1 - e] = c0; }
        __syncthreads();
    }
    o.x_s = xs; o.x_e = xe; o.y_s = ys; o.y_e = ye; o.rev = (uint8_t)rev; o.score = score_best; o.n_chain = cnt; o.valid = 1;
    o.n_win = xe / FSV_WINDOW - xs / FSV_WINDOW + 1;
    // mirror: same anchors seen from t; on the reverse strand both coordinates are measured from the other read end
    om.rev = (uint8_t)rev; om.score = o.score; om.n_chain = cnt; om.valid = 1;
    if (!rev) { om.x_s = ys; om.x_e = ye; om.y_s = xs; om.y_e = xe; }
    else { om.x_s = lent - 1 - ye; om.x_e = lent - 1 - ys; om.y_s = lenq - 1 - xe; om.y_e = lenq - 1 - xs; }
    om.n_win = om.x_e / FSV_WINDOW - om.x_s / FSV_WINDOW + 1;
    if (!A.emit_tasks) { o.n_win = 0; om.n_win = 0; PUT_BOTH(); return; }
    // 7. window tasks of both directions
    uint32_t first_win = 0;
    if (lane == 0) {
        first_win = atomicAdd(A.task_counter, (uint32_t)(o.n_win + om.n_win));
        // statistics: DP columns of the windows handed to K5, both directions; one counter per set (indexed by the set's first
        // read) -- a single shared counter costs ~10 ns per pair in same-address atomics
        atomicAdd(&A.set_cols[pt.x], (uint32_t)(xe - xs + 1) + (uint32_t)(om.x_e - om.x_s + 1));
    }
    first_win = __shfl(first_win, 0, 64);
    if ((uint64_t)first_win + (uint32_t)(o.n_win + om.n_win) > A.task_cap) {
        if (lane == 0) { atomicExch(A.overflow, 1u); o.valid = 0; o.n_win = 0; om.valid = 0; om.n_win = 0; A.ovl[p] = o; A.ovl[pm] = om; }
        return;
    }
    o.first_win = (int32_t)first_win;
    om.first_win = (int32_t)(first_win + (uint32_t)o.n_win);
    const uint32_t xw = A.word_off[rq], yw = A.word_off[rt];
    // chain anchor e (start-to-end order, 0 <= e < cnt)
#define CH_Q(e) KEY_Q(s_chain[cnt - 1 - (e)])
#define CH_T(e) KEY_T(s_chain[cnt - 1 - (e)])
    {
        const int w0 = xs / FSV_WINDOW;
        for (int j = lane; j < o.n_win; j += 64) {
            const int gs = (w0 + j) * FSV_WINDOW, ge = gs + FSV_WINDOW - 1;
            const int x_start = max(gs, xs);
            const int x_len = min(ge, xe) - x_start + 1;
            // diagonal of the last chain anchor with qe <= x_start, else of the first one
            int lo2 = 0, hi2 = cnt;
            while (lo2 < hi2) { int mid = (lo2 + hi2) >> 1; if (CH_Q(mid) <= x_start) lo2 = mid + 1; else hi2 = mid; }
            const int e = lo2 == 0 ? 0 : lo2 - 1;
            const int diag = CH_T(e) - CH_Q(e);
            fsv_wtask w;
            w.x_word = xw; w.y_word = yw; w.x_start = x_start; w.y_start = x_start + diag; w.y_len = lent;
            w.x_len = (uint16_t)x_len; w.k = A.thr_tab[x_len]; w.y_rev = (uint8_t)rev; w.ovl = p; w.win = (uint32_t)j;
            A.tasks[first_win + j] = w;
        }
    }
    {
        // mirrored direction: query t, target q.  Mirrored anchor of e: same strand (ct_e, cq_e) in the same order;
        // reverse strand (lent-1-ct_e, lenq-1-cq_e) in reversed order.
        const int mxs = om.x_s, mxe = om.x_e, w0 = mxs / FSV_WINDOW;
        for (int j = lane; j < om.n_win; j += 64) {
            const int gs = (w0 + j) * FSV_WINDOW, ge = gs + FSV_WINDOW - 1;
            const int x_start = max(gs, mxs);
            const int x_len = min(ge, mxe) - x_start + 1;
            int diag;
            if (!rev) {
                int lo2 = 0, hi2 = cnt; // last anchor with ct <= x_start
                while (lo2 < hi2) { int mid = (lo2 + hi2) >> 1; if (CH_T(mid) <= x_start) lo2 = mid + 1; else hi2 = mid; }
                const int e = lo2 == 0 ? 0 : lo2 - 1;
                diag = CH_Q(e) - CH_T(e);
            } else {
                // mirrored query coordinate lent-1-ct_e decreases with e: the last mirrored anchor with coordinate <= x_start is the
                // smallest e with ct_e >= lent-1-x_start; none -> the first mirrored anchor (e = cnt-1)
                const int thr = lent - 1 - x_start;
                int lo2 = 0, hi2 = cnt; // first e with ct_e >= thr
                while (lo2 < hi2) { int mid = (lo2 + hi2) >> 1; if (CH_T(mid) < thr) lo2 = mid + 1; else hi2 = mid; }
                const int e = lo2 == cnt ? cnt - 1 : lo2;
                diag = (lenq - 1 - CH_Q(e)) - (lent - 1 - CH_T(e));
            }
            fsv_wtask w;
            w.x_word = yw; w.y_word = xw; w.x_start = x_start; w.y_start = x_start + diag; w.y_len = lenq;
            w.x_len = (uint16_t)x_len; w.k = A.thr_tab[x_len]; w.y_rev = (uint8_t)rev; w.ovl = pm; w.win = (uint32_t)j;
            A.tasks[om.first_win + j] = w;
        }
    }
#undef CH_Q
#undef CH_T
    PUT_BOTH();
    CH_MARK(6);
#undef CH_MARK
#undef PUT_BOTH
#undef KEY_Q
#undef KEY_T
#undef MAKE_KEY
}

template <bool SHORT>
__device__ __forceinline__ void chain_load_query(uint4 (&qa)[FSV_CHAIN_QR], const fsv_mz *mq, int nq)
{
    const uint4 *mq4 = (const uint4 *)mq;
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));   // (see chain_pair)
    if (nq <= 64 * FSV_CHAIN_QR) {
#pragma unroll
        for (int u = 0; u < FSV_CHAIN_QR; u++) { const int i = u * 64 + lane; qa[u] = i < nq ? mq4[i] : make_uint4(0, 0, 0, 0); }
    }
}

// one block per listed pair (the re-chaining of a few pairs with another band width)
template <bool SHORT>
__global__ __launch_bounds__(64) void k_chain(ChainArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    if (A.pair_list && A.n_list_dev && blockIdx.x >= *A.n_list_dev) return;
    const uint4 pt = A.upair_tab[A.pair_list ? A.pair_list[blockIdx.x] : blockIdx.x];
    const uint32_t rq = pt.x + (pt.y & 0xffffu), rt = pt.x + (pt.y >> 16);
    const int lenq = A.read_len[rq], lent = A.read_len[rt];
    const int nq = (int)A.mz_cnt[rq], nt = (int)A.mz_cnt[rt];
    const fsv_mz *mq = A.mz + A.mz_off[rq] + nq, *mt = A.mz + A.mz_off[rt]; // q: position-sorted copy, t: hash-sorted
    if (A.wide_list && min(nq, nt) > A.amax) {
        if (threadIdx.x == 0) A.wide_list[atomicAdd(A.n_wide, 1u)] = A.pair_list ? A.pair_list[blockIdx.x] : blockIdx.x;
        return;
    }
    uint4 qa[FSV_CHAIN_QR];
    chain_load_query<SHORT>(qa, mq, nq);
    chain_pair<SHORT>(A, s_raw, pt, lenq, lent, nq, nt, mq, mt, qa);
}

// the pairs set aside by the kernels above (long reads: more than 1 024 minimizers in both lists), with the large tile; a small
// fixed grid walks the list, which is empty for reads below ~25 kb
template <bool SHORT>
__global__ __launch_bounds__(64) void k_chain_wide_list(ChainArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const uint32_t n = *A.n_wide;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint4 pt = A.upair_tab[A.wide_list[i]];
        const uint32_t rq = pt.x + (pt.y & 0xffffu), rt = pt.x + (pt.y >> 16);
        const int lenq = A.read_len[rq], lent = A.read_len[rt];
        const int nq = (int)A.mz_cnt[rq], nt = (int)A.mz_cnt[rt];
        const fsv_mz *mq = A.mz + A.mz_off[rq] + nq, *mt = A.mz + A.mz_off[rt];
        uint4 qa[FSV_CHAIN_QR];
        chain_load_query<SHORT>(qa, mq, nq);
        chain_pair<SHORT>(A, s_raw, pt, lenq, lent, nq, nt, mq, mt, qa);
        __syncthreads();
    }
}

// The full pass: one block per FSV_CHAIN_CH consecutive pairs of the pair table (row-major: the pairs of a chunk nearly always
// share their query).  A block per pair paid three dependent memory round trips before its first lookup (pair record -> lengths /
// counts / offsets -> the two lists: two thirds of the kernel's wave cycles, FSV_CHAIN_STAMPS); here the chunk's records and
// its targets' lengths, counts and offsets arrive in two trips for all its pairs, and a pair waits for one round trip -- the two
// lists.  (One block per whole row was slower: rows have 0 .. ns-1 pairs.)
#define FSV_CHAIN_CH 8
template <bool SHORT>
__global__ __launch_bounds__(64) void k_chain_chunks(ChainArgs A, uint32_t n_upairs)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int lane = threadIdx.x;
    uint32_t chunk;
    if (!xcd_block((n_upairs + FSV_CHAIN_CH - 1) / FSV_CHAIN_CH, chunk)) return;
    const uint32_t first = chunk * FSV_CHAIN_CH;
    const int np = (int)min((uint32_t)FSV_CHAIN_CH, n_upairs - first);
    uint4 ptv = make_uint4(0, 0, 0, 0);
    int lent_v = 0, nt_v = 0;
    uint32_t offt_v = 0;
    if (lane < np) {
        ptv = A.upair_tab[first + lane];
        const uint32_t rt = ptv.x + (ptv.y >> 16);
        lent_v = A.read_len[rt]; nt_v = (int)A.mz_cnt[rt]; offt_v = A.mz_off[rt];
    }
    // q's list stays in registers while the query does not change (half the list traffic; FSV_CHAIN_QR keeps the kernel at three
    // waves per SIMD with it)
    uint32_t cur_rq = 0xffffffffu;
    int lenq = 0, nq = 0;
    uint32_t offq = 0;
    uint4 qa[FSV_CHAIN_QR];
#pragma nounroll
    for (int i = 0; i < np; i++) {
        // readlane: the pair's values are wave-uniform and must live in scalar registers
#define RL(v_) ((uint32_t)__builtin_amdgcn_readlane((int)(v_), i))
        const uint4 pt = make_uint4(RL(ptv.x), RL(ptv.y), RL(ptv.z), RL(ptv.w));
        const int lent = (int)RL(lent_v), nt = (int)RL(nt_v);
        const uint32_t offt = RL(offt_v);
#undef RL
        const uint32_t rq = pt.x + (pt.y & 0xffffu);
        if (rq != cur_rq) {
            cur_rq = rq;
            lenq = __builtin_amdgcn_readfirstlane(A.read_len[rq]); nq = __builtin_amdgcn_readfirstlane((int)A.mz_cnt[rq]);
            offq = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.mz_off[rq]);
            chain_load_query<SHORT>(qa, A.mz + offq + nq, nq);   // the position-sorted copy
        }
        if (A.wide_list && min(nq, nt) > A.amax) {     // only such a pair can have more anchors than this tile holds
            if (lane == 0) A.wide_list[atomicAdd(A.n_wide, 1u)] = first + (uint32_t)i;
            continue;
        }
        const fsv_mz *mq = A.mz + offq + nq;
        chain_pair<SHORT>(A, s_raw, pt, lenq, lent, nq, nt, mq, A.mz + offt, qa);
        __syncthreads();   // the next pair reuses the tile
    }
}

// LDS bytes of a k_chain block
__host__ __device__ inline size_t chain_lds_bytes(bool short_reads, int amax) { return short_reads ? (size_t)amax * 12 + (size_t)amax / 8 : (size_t)amax * 24; }

// ------------------------------------------------------------------------------------------------ k_rescue_accept
// One lane per overlap slot: right-extension rescue of unmatched windows (Correct.cpp:2655-2744),
// then the 0.9 coverage filter and the 0.03 error-rate filter (Correct.cpp:2899-3021, 725).
__device__ __forceinline__ int double_thr(int pre, int x_len, int k_cap)
{
    if (pre == 0 && x_len >= 4) pre = 1;
    int t = pre * 2;
    if (x_len >= 300 && t < k_cap) t = k_cap;
    if (t > k_cap) t = k_cap;
    return t;
}

// WIDE: the batch's error model allows thresholds above 31 (k_cap up to 95): the re-runs go through the wide-band BPM
template <bool WIDE>
__global__ __launch_bounds__(64) void k_rescue_accept(const uint32_t *__restrict__ store, fsv_ovl *__restrict__ ovl, uint32_t n_pairs,
                                                      fsv_wtask *__restrict__ tasks, fsv_wres *__restrict__ res,
                                                      unsigned long long *__restrict__ stat_cols, uint4 *__restrict__ ovl_c, int k_cap, int accept_err_pm,
                                                      uint32_t *__restrict__ left_list, uint32_t *__restrict__ n_left)
{
    const uint32_t p = blockIdx.x * 64 + threadIdx.x;
    if (p >= n_pairs) return;
    fsv_ovl o = ovl[p];
    // ovl_c: what the consensus needs of an overlap, 16 B instead of 56: {x_s, first window task, n_win | accepted << 31, -}
    if (!o.valid) { ovl_c[p] = make_uint4(0u, 0u, 0u, 0u); return; }
    fsv_wtask *T = tasks + o.first_win;
    fsv_wres *R = res + o.first_win;
    int align = 0;
    unsigned long long cols = 0;
    // one pass settles an overlap whose windows all matched (nearly all of them): aligned length, total length and error sum;
    // four windows per trip so that their loads are in flight together (a lane walks ~20 windows, a memory round trip each)
    int n_bad = 0;
    long long tlen0 = 0, terr0 = 0;
#pragma unroll 4
    for (int j = 0; j < o.n_win; j++) {
        const int e = R[j].err, xl = T[j].x_len;
        if (e >= 0) { align += xl; terr0 += e; } else n_bad++;
        tlen0 += xl;
    }
    for (int j = n_bad ? o.n_win - 1 : -1; j >= 0; j--) {
        if (R[j].err < 0) continue;
        int next = R[j].y_beg + R[j].end_site - R[j].extra_begin + 1;
        for (int k2 = j + 1; k2 < o.n_win && R[k2].err < 0; k2++) {
            fsv_wtask u = T[k2];
            if (next >= u.y_len) break;
            u.k = (uint8_t)double_thr(u.k, u.x_len, k_cap);
            u.y_start = next;
            fsv_wres r;
            if (!bpm_window_geometry(u, r, k_cap)) break;
            if ((u.x_len + 2 * u.k - r.extra_begin - r.extra_end) + u.k < u.x_len) break;
            if (WIDE) { WideNoSink none; bpm_run_wide(store, u, r, none, k_cap); }
            else bpm_run(store, u, r, BpmNoSink());
            cols += u.x_len;
            if (r.err < 0) break;
            T[k2] = u; R[k2] = r;
            align += u.x_len;
            next = r.y_beg + r.end_site - r.extra_begin + 1;
        }
    }
    if (n_bad && left_list) {
        // an unmatched window left of a matched one: the left-extension pass (k_left_rescue) decides about this overlap
        bool left = false;
        for (int j = 1; j < o.n_win && !left; j++) left = R[j].err >= 0 && R[j - 1].err < 0;
        if (left) {
            left_list[atomicAdd(n_left, 1u)] = p;
            o.is_match = 0; ovl[p] = o;
            ovl_c[p] = make_uint4((uint32_t)o.x_s, (uint32_t)o.first_win, (uint32_t)o.n_win, 0u);
            if (cols) atomicAdd(stat_cols, cols);
            return;
        }
    }
    long long tlen = tlen0, terr = terr0;
    if (n_bad) {   // the rescue may have changed results and window lengths never change: only the error sum is taken again
        terr = 0;
#pragma unroll 4
        for (int j = 0; j < o.n_win; j++) { const int e = R[j].err; terr += e >= 0 ? e : T[j].x_len; }
    }
    o.align_len = align; o.err_sum = (int32_t)terr;
    o.is_match = ((long long)(o.x_e - o.x_s + 1) * 9 <= (long long)align * 10 && terr * 1000 <= tlen * accept_err_pm) ? 1 : 0;
    ovl[p] = o;
    ovl_c[p] = make_uint4((uint32_t)o.x_s, (uint32_t)o.first_win, (uint32_t)o.n_win | (o.is_match ? 0x80000000u : 0u), 0u);
    if (cols) atomicAdd(stat_cols, cols);
}

// ------------------------------------------------------------------------------------------------ K6 paths
__device__ __forceinline__ void ops_set(uint8_t *ops, int i, uint32_t v) { ops[i >> 2] = (uint8_t)((ops[i >> 2] & ~(3u << ((i & 3) << 1))) | (v << ((i & 3) << 1))); }
__device__ __forceinline__ uint32_t ops_get(const uint8_t *ops, int i) { return (ops[i >> 2] >> ((i & 3) << 1)) & 3u; }

// y base of the padded window column c for a task
__device__ __forceinline__ uint32_t task_ybase(const uint32_t *__restrict__ store, const fsv_wtask &t, int c)
{
    return bpm_ywin_base(store, t, t.y_start - t.k, c);
}

// Fast paths of Reserve_Banded_BPM_PATH (Levenshtein_distance.h:516-531): err == 0, or a gap-free placement with
// exactly err mismatches (try_cigar).  Everything else is queued for one of the walk kernels.
struct PathLists {      // task lists and their device-side lengths: 0-2 k_path_fr<1..3>, 3 k_path_sb, 4 k_path_dp<32>, 5 k_path_dp<64>, 6 k_path_wide,
    uint32_t *list[8], *cnt[8];   // 7 (may be null): windows whose alignment may touch the edge of its band (k_fix_boundary looks at them again)
};
// try_cigar (Levenshtein_distance.h:465-507): the gap-free placement on the end diagonal K5 reported.  When its mismatches are the
// window's distance that is the path (generate_cigar then only trims mismatches at the two ends into x-only ops): the record is
// written and true returned; false: the window needs a walk.  One lane per window, no cross-lane traffic.
__device__ __forceinline__ bool path_gapfree(const uint32_t *__restrict__ store, const fsv_wtask &t, const fsv_wres &r, fsv_wpath *__restrict__ P, bool write_clean_ops)
{
    const int n = t.x_len;
    const int start = r.end_site - n + 1;
    // mismatch map of the gap-free placement, 16 columns per word; field value 1 == op "mismatch"
    uint32_t ops32[26];
#pragma unroll
    for (int i = 0; i < 26; i++) ops32[i] = 0;
    bool ok = r.err == 0;
    if (!ok && start >= 0) {
        int mm = 0;
        const int win0 = t.y_start - t.k;
#pragma unroll
        for (int c = 0; c < 6; c++) {
            if (c * 64 < n) {
                uint32_t xb4[4], yb4[4], yv4[4];
                fetch64_x(store, t.x_word, t.x_start + c * 64, xb4);
                fetch64(store, t.y_word, t.y_len, t.y_rev, win0 + start + c * 64, yb4, yv4);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int b = c * 4 + q;
                    if (b * 16 < n) {
                        uint32_t d = xb4[q] ^ yb4[q];
                        d = (d | (d >> 1)) & 0x55555555u;
                        // columns outside the read ('N') never match; columns past the window do not count
                        uint32_t inval = ~yv4[q] & 0xffffu, spread = 0;
                        for (int j = 0; j < 16; j++) spread |= ((inval >> j) & 1u) << (2 * j);
                        d |= spread;
                        const int lim = min(16, n - b * 16);
                        if (lim < 16) d &= (1u << (2 * lim)) - 1u;
                        ops32[b] = d;
                        mm += __popc(d);
                    }
                }
            }
        }
        ok = (mm == r.err);
    }
    if (!ok) return false;
    // gap-free path.  generate_cigar (Correct.cpp:1387-1536) turns mismatches at either end into x-only ops (3) and
    // moves the y interval inwards -- the alignment end first, then its start; there are no gaps to shift.
    int s2 = start, e2 = r.end_site;
    if (r.err > 0) {
        for (int i = n - 1; i >= 0 && ((ops32[i >> 4] >> ((i & 15) << 1)) & 3u) == 1u; i--) { ops32[i >> 4] |= 3u << ((i & 15) << 1); e2--; }
        for (int i = 0; i < n && ((ops32[i >> 4] >> ((i & 15) << 1)) & 3u) == 1u; i++) { ops32[i >> 4] |= 3u << ((i & 15) << 1); s2++; }
    }
    P->ry_start = t.y_start - t.k + s2;
    P->ry_end = t.y_start - t.k + e2;
    uint32_t flags10 = 0;      // as in path_finish: an op other than a match among the first / last ten
    if (r.err > 0) {
        const int a = max(n - 10, 0), wi = a >> 4, sh = (a & 15) << 1;
        const uint32_t lo = ops32[wi] >> sh, hi = (sh && wi + 1 < 26) ? ops32[wi + 1] << (32 - sh) : 0u;
        flags10 = ((ops32[0] & 0xfffffu) ? 1u : 0u) | (((lo | hi) & 0xfffffu) ? 2u : 0u) | (start == 0 ? 4u : 0u);   // bit 2: the alignment starts in the padded window's first column
    }
    P->path_len = (int16_t)n; P->err = (int16_t)r.err; P->state = 1; P->y_rev = t.y_rev; P->pad = (uint16_t)flags10; P->y_word = t.y_word; P->y_len = t.y_len;
    // a distance-0 record carries no ops: its consumers (k_consensus, k_het) look at err first and never read them, and in the
    // later correction rounds nearly every window is one -- 24 bytes out instead of 128
    if (r.err == 0 && !write_clean_ops) return true;
    uint2 *dst = reinterpret_cast<uint2 *>(P->ops); // ops sit at byte 24 of the record: 8-byte aligned
#pragma unroll
    for (int i = 0; i < 13; i++) dst[i] = make_uint2(ops32[2 * i], ops32[2 * i + 1]);
    return true;
}

__global__ __launch_bounds__(256) void k_path_fast(const uint32_t *__restrict__ store, const fsv_ovl *__restrict__ ovl,
                                                   const fsv_wtask *__restrict__ tasks, const fsv_wres *__restrict__ res, uint32_t n_tasks,
                                                   fsv_wpath *__restrict__ paths, PathLists L, bool write_clean_ops, const uint32_t *__restrict__ n_dev)
{
    if (n_dev) n_tasks = min(*n_dev, n_tasks);   // the grid covers the task bound; the count stays on the device (no host round trip), clamped to the bound
    uint32_t blk;
    if (!xcd_block((n_tasks + 255u) >> 8, blk)) return;
    const uint32_t tid = blk * blockDim.x + threadIdx.x;
    if (tid >= n_tasks) return;
    const fsv_wtask t = tasks[tid];
    const fsv_wres r = res[tid];
    fsv_wpath *P = paths + tid;
    if (r.err < 0 || !ovl[t.ovl].is_match) { P->state = 0; return; }
    const bool ok = path_gapfree(store, t, r, P, write_clean_ops);
    // fix_boundary's candidates, from what K5 knows: the alignment ends in the padded window's last column, or could start in its first
    // one (it starts at end - (n - 1) - (inserted - deleted bases), and the distance bounds that difference)
    if (L.cnt[7] && r.err > 0 && t.k <= FSV_K_MAX && (r.end_site == (int)t.x_len + 2 * (int)t.k - 1 || r.end_site - ((int)t.x_len - 1) <= r.err))
        L.list[7][atomicAdd(L.cnt[7], 1u)] = tid;
    // Not settled here: queued for one of the walk kernels, each list homogeneous -- first-pass bands (k <= 15) by distance: the walk
    // without the matrix up to 3 (nine in ten; a list per distance), the sub-band matrix up to 7, the general kernel beyond; the doubled thresholds of the
    // rescue pass (k <= 31); bands above 63 rows (k_path_wide).  One atomic instruction per wave: a class's first lane reserves its slots.
    {
        const int cls = ok ? -1 : (t.k <= FSV_K_MAX && r.err <= FSV_FR_MAXERR) ? r.err - 1 : t.k <= 15 ? (r.err <= FSV_SB_MAXERR ? 3 : 4) : t.k <= FSV_K_MAX ? 5 : 6;
        if (__any(cls >= 0)) {
            const int lane = (int)(threadIdx.x & 63u);
            unsigned long long mine = 0ull;
#pragma unroll
            for (int c = 0; c < 7; c++) {
                const unsigned long long m = __ballot(cls == c);
                if (cls == c) mine = m;
            }
            const int leader = mine ? __ffsll((long long)mine) - 1 : lane;      // the class's first lane reserves for all of them
            uint32_t base = 0, *cnt = L.cnt[0], *list = L.list[0];
#pragma unroll
            for (int c = 1; c < 7; c++) { cnt = cls == c ? L.cnt[c] : cnt; list = cls == c ? L.list[c] : list; }
            if (cls >= 0 && lane == leader) base = atomicAdd(cnt, (uint32_t)__popcll(mine));
            const uint32_t at = __shfl(base, leader, 64) + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
            if (cls >= 0) { P->state = 2; list[at] = tid; }
        }
    }
}

// base access through one cached 16-base word (forward position >> 4 is the key)
struct XBaseCache {
    uint32_t w = 0; int idx = -1;
    __device__ __forceinline__ uint32_t get(const uint32_t *__restrict__ store, uint32_t word_off, int pos)
    {
        const int wi = pos >> 4;
        if (wi != idx) { w = store[word_off + (uint32_t)wi]; idx = wi; }
        return (w >> ((pos & 15) << 1)) & 3u;
    }
};
struct YBaseCache {   // task_ybase: window column c of the (strand-oriented) target, 4 outside the read
    uint32_t w = 0; int idx = -1;
    __device__ __forceinline__ uint32_t get(const uint32_t *__restrict__ store, const fsv_wtask &t, int c)
    {
        const int p = t.y_start - t.k + c;
        if (p < 0 || p >= t.y_len) return 4u;
        const int q = t.y_rev ? (t.y_len - 1 - p) : p, wi = q >> 4;
        if (wi != idx) { w = store[t.y_word + (uint32_t)wi]; idx = wi; }
        const uint32_t b = (w >> ((q & 15) << 1)) & 3u;
        return t.y_rev ? 3u - b : b;
    }
};

// The end of K6, shared by the two DP kernels: the all-match rest of the walk, generate_cigar's end trimming and greedy gap
// left-shift (Correct.cpp:1302-1536) on the path in LDS (2 bits per op, end-to-start), and the record.
#define TMP(i) ((s_ops[(i) >> 4][lane64] >> (((i) & 15) << 1)) & 3u)
#define TMP_SET(i, v) do { const int w_ = (i) >> 4, sh_ = ((i) & 15) << 1; s_ops[w_][lane64] = (s_ops[w_][lane64] & ~(3u << sh_)) | ((uint32_t)(v) << sh_); } while (0)
__device__ __forceinline__ void path_finish(const uint32_t *__restrict__ store, const fsv_wtask &t, fsv_wpath *__restrict__ P, uint32_t (*s_ops)[64],
                                            int lane64, int col, int dir, int plen, int start, int end, int err)
{
    if (col > 0) { start -= col; plen += col; dir = 0; } // the rest of the path is matches: the fields are already 0
    // a record holds FSV_PATH_CAP ops (x_len + k <= 406 for hifiasm's thresholds: never reached); a longer path -- a wide-band
    // window with more than 41 inserted bases -- leaves the window without a path, as oracle/asm.c:window_path does
    if (plen > FSV_PATH_CAP) { P->state = 0; return; }
    if (dir != 3) start++;
    const uint32_t raw0 = (err > 0 && start == 0) ? 4u : 0u;      // the alignment starts in the padded window's first column (before generate_cigar moves it): fix_boundary's question
    // generate_cigar: TMP is stored end-to-start
    if (err > 0) {
        int stop = -1;
        for (int i = 0; i < plen && TMP(i) == 1; i++) { TMP_SET(i, 3); end--; stop = i; }
        for (int i = plen - 1; i >= 0 && TMP(i) == 1; i--) { TMP_SET(i, 3); start++; }
        int xi = 0, yi = 0;
        XBaseCache xc; YBaseCache yc;
        for (int i = plen - 1; i > stop;) {
            // runs of match / mismatch ops are skipped a path word at a time: a gap op is a field with its high bit set
            const int f = i & 15;
            const uint32_t wv = s_ops[i >> 4][lane64];
            const uint32_t m = wv & 0xAAAAAAAAu & (f == 15 ? 0xffffffffu : ((1u << (2 * f + 2)) - 1u));
            const int skip = min(m == 0u ? f + 1 : f - ((31 - __clz(m)) >> 1), i - stop);
            if (skip > 0) { xi += skip; yi += skip; i -= skip; continue; }
            const uint32_t op = (wv >> (2 * f)) & 3u;
            // shift this gap towards the alignment start while the bases it passes still pair up (move_gap_greedy)
            int pi = i + 1, x2 = xi, y2 = yi;
            if (op == 3) y2--; else x2--;
            for (; pi < plen && x2 >= 0 && y2 >= 0; pi++, x2--, y2--) {
                const uint32_t pv = TMP(pi);
                // the shift reads bases at falling positions: one 16-base word per 16 steps instead of two dependent global loads
                // per step (a gap inside a homopolymer travels a long way, and the whole wave waits for its slowest lane)
                const bool same = xc.get(store, t.x_word, t.x_start + x2) == yc.get(store, t, start + y2);
                if (pv >= 2 || (pv == 0 && !same)) break;
                if (pv == 1 && same) { TMP_SET(pi - 1, 0); err--; }
                else TMP_SET(pi - 1, pv);
                TMP_SET(pi, op);
            }
            if (op == 2) yi++; else xi++;
            i--;
        }
    }
    // pack start-to-end: output word wd holds the source fields plen-16-16wd .. plen-1-16wd in reverse order
    const int pl = min(plen, FSV_PATH_CAP);
    uint32_t *dst = reinterpret_cast<uint32_t *>(P->ops);
    uint32_t head10 = 0;
    for (int wd = 0; wd < 26; wd++) {
        const int a = plen - 16 - 16 * wd;
        uint32_t v = 0;
        if (a >= 0) {
            const int wi = a >> 4, sh = (a & 15) << 1;
            const uint32_t w0 = s_ops[wi][lane64], w1 = (sh && wi + 1 < 28) ? s_ops[wi + 1][lane64] : 0u;
            v = sh ? (w0 >> sh) | (w1 << (32 - sh)) : w0;
        } else if (a > -16) v = s_ops[0][lane64] << ((-a) << 1);
        v = rev_fields2(v);
        if (wd == 0) head10 = v & 0xfffffu;
        dst[wd] = v;
    }
    P->ry_start = t.y_start - t.k + start;
    P->ry_end = t.y_start - t.k + end;
    // pad: bit 0 = an op other than a match among the first ten, bit 1 = among the last ten -- what scan_cigar (Correct.cpp:1070) over ten
    // columns from either end asks about (calculate_boundary_cigars :2360; k_bcig_tasks reads the header only)
    const uint32_t tail10 = s_ops[0][lane64] & 0xfffffu;
    P->path_len = (int16_t)pl; P->err = (int16_t)err; P->state = 1; P->y_rev = t.y_rev; P->pad = (uint16_t)((head10 ? 1u : 0u) | (tail10 ? 2u : 0u) | raw0); P->y_word = t.y_word; P->y_len = t.y_len;
}
#undef TMP
#undef TMP_SET

// General K6 (any band up to 63 rows, any distance): forward pass keeping {D0, VP, VN} of every column in a per-lane slice of
// an HBM scratch ([block][column][word][lane]), the reference's walk back on those words, then path_finish.  Since round 2
// this is the fallback: first-pass windows (k <= 15) at distance <= FSV_SB_MAXERR go through k_path_sb below, which keeps
// 4 bytes per column instead of 12 and walks without a dependent global load per step; what is left for this kernel are
// the doubled-threshold rescue windows (k > 15) and distances above 7 -- a fraction of a percent of the HiFi windows.
// WordT = uint32_t for bands of at most 31 diagonals (k <= 15): the walk back only looks at bits below the band width, so the
// low halves of D0 / VP / VN are all it needs and the scratch traffic halves; uint64_t for the doubled thresholds (k <= 31).
template <class WordT> struct PathSink {
    WordT *cols; uint32_t stride, lane;
    __device__ __forceinline__ void operator()(int i, uint64_t d0, uint64_t vp, uint64_t vn) const
    {
        WordT *c = cols + (size_t)(i + 1) * 3 * stride + lane;
        c[0] = (WordT)d0; c[stride] = (WordT)vp; c[2 * (size_t)stride] = (WordT)vn;
    }
};

// Reserve_Banded_BPM_PATH by one lane for one window: the forward pass with every column's {D0, VP, VN} kept in the lane's slice of the
// HBM scratch, the walk back, generate_cigar and the record (path_finish).  The general K6 kernel's body; k_left_rescue calls it too.
template <class WordT>
__device__ __forceinline__ void path_general(const uint32_t *__restrict__ store, const fsv_wtask &t, fsv_wpath *__restrict__ P, uint32_t (*s_ops)[64],
                                             int lane64, WordT *cols_slice, uint32_t cstride = 64u)
{
    // cols_slice: the scratch of this lane's block, word w of column c of lane l at ((c) * 3 + w) * cstride + l (k_path_dp: 64 lanes a
    // block in HBM; k_left_rescue: one lane a block, in LDS)
    const int n = t.x_len, k = t.k, band = 2 * k + 1;
    fsv_wres r;
    PathSink<WordT> sink{cols_slice, cstride, cstride == 1u ? 0u : (uint32_t)lane64};
    bpm_run(store, t, r, sink);
    if (r.err < 0) { P->state = 0; return; } // cannot happen: K5 matched this window
#define COL(c, w) (sink.cols[((c) * 3 + (w)) * (size_t)sink.stride + sink.lane])
    for (int i = 0; i < 28; i++) s_ops[i][lane64] = 0;
    int end = r.end_site, err = r.err;
    int cur = err, col = n, plen = 0, start = end, row = band - (n + 2 * k - end), dir = 0;
    {
    // the kernel is instruction-bound (4-5 waves per SIMD keep the issue slots full), so the walk is written for few
    // instructions: WordT-wide bit tests (only band bits are read), the column it leaves behind handed to the next step
    // instead of re-read, and the ops gathered in a register that goes to LDS once per 16 steps
    WordT vp = COL(col, 1), vn = COL(col, 2);
    uint32_t acc = 0;
    while (col > 0 && cur != 0) {
        const WordT d0 = COL(col, 0);
        const WordT vpi = col > 1 ? COL(col - 1, 1) : (WordT)0, vni = col > 1 ? COL(col - 1, 2) : (WordT)0;
        const WordT hn = vpi & d0, hp = vni | ~(vpi | d0);
        const int diag = cur - (int)((~(d0 >> row)) & 1u);
        const bool can_up = row != 0, can_left = row == 0 || row != band - 1;
        int left = cur, up = cur;
        if (can_left) left = cur - (int)((hp >> row) & 1u) + (int)((hn >> row) & 1u);
        if (can_up) up = cur - (int)((vp >> (row - 1)) & 1u) + (int)((vn >> (row - 1)) & 1u);
        int best = diag; dir = 0;
        if (can_up && up < best) { best = up; dir = 2; }
        if (can_left && left < best) { best = left; dir = 3; }
        if (dir == 0) { if (diag != cur) dir = 1; col--; start--; vp = vpi; vn = vni; }
        else if (dir == 2) { row--; start--; }
        else { col--; row++; vp = vpi; vn = vni; }
        acc |= (uint32_t)dir << ((plen & 15) << 1);
        if ((plen & 15) == 15) { s_ops[plen >> 4][lane64] = acc; acc = 0; }
        plen++;
        cur = best;
    }
    if (plen & 15) s_ops[plen >> 4][lane64] = acc;
    }
    path_finish(store, t, P, s_ops, lane64, col, dir, plen, start, end, err);
#undef COL
}

template <class WordT>
__global__ __launch_bounds__(64) void k_path_dp(const uint32_t *__restrict__ store, const fsv_wtask *__restrict__ tasks,
                                                const uint32_t *__restrict__ dp_list, uint32_t list_begin, uint32_t list_end,
                                                fsv_wpath *__restrict__ paths, WordT *__restrict__ cols, uint32_t stride,
                                                const uint32_t *__restrict__ n_dev)
{
    __shared__ uint32_t s_ops[28][64];     // per lane: the path being built, 2 bits per op, stored end-to-start (448 ops)
    const int lane64 = threadIdx.x;
    if (n_dev) list_end = list_begin + *n_dev;   // the list's length as the kernel before left it: no host round trip
    const uint32_t slot = blockIdx.x * 64 + threadIdx.x;   // this lane's slice of the scratch, reused for every task it takes
    // persistent blocks: the grid is sized to what the device holds at once and every block strides through the list, so the
    // scratch is a few hundred MB whatever the number of windows, and the whole list is one launch
    for (uint32_t li = list_begin + slot; li < list_end; li += gridDim.x * 64) {
        const uint32_t tid = dp_list[li];
        const fsv_wtask t = tasks[tid];
        path_general<WordT>(store, t, paths + tid, s_ops, lane64, cols + (size_t)blockIdx.x * (FSV_WINDOW + 2) * 3 * 64);
    }
    (void)stride;
}

// ------------------------------------------------------------------------------------------------ fix_boundary
// fix_boundary (Correct.cpp:1676-1795; for the windows' final cigars :2968 and in the left-extension pass :2858): an alignment that
// starts in the first column of its padded window, or ends in its last one, may have been cut off by the band -- the window is aligned
// once more with the band shifted by k towards that side (from the old region's first base / so that the x interval ends at the old
// alignment's last base), without a hint, and the new alignment stands when it has fewer errors.  t / r / the record at P: the window
// as K6 left it; they are replaced when the new alignment stands.  One lane, column scratch `cols` with lane stride 1 (LDS).
// oracle/asm.c:window_path is the same, statement for statement.
__device__ __forceinline__ void fix_boundary_dev(const uint32_t *__restrict__ store, fsv_wtask &t, fsv_wres &r, fsv_wpath *__restrict__ P, uint32_t (*s_ops)[64],
                                                 uint64_t *cols, int k_cap)
{
    const uint4 h = *reinterpret_cast<const uint4 *>(P);
    if ((h.w & 0xffu) != 1u || r.err <= 0 || t.k > FSV_K_MAX || (r.extra_begin & 0x4000)) return;     // (bit 14 of extra_begin: moved once already)
    const int n = t.x_len, k = t.k, wlen = n + 2 * k;
    const bool raw0 = ((h.w >> 16) & 4u) != 0u;
    fsv_wtask t2 = t;
    if (raw0) { if (r.extra_begin != 0) return; t2.y_start = r.y_beg; }
    else if (r.end_site == wlen - 1) { if (r.extra_end != 0) return; t2.y_start = (r.y_beg + r.end_site) - n + 1; }
    else return;
    fsv_wres r2;
    if (!bpm_window_geometry(t2, r2, k_cap)) return;
    if (r2.y_beg == r.y_beg) return;
    bpm_run(store, t2, r2, BpmNoSink());
    if (r2.err < 0 || r2.err >= r.err) return;
    path_general<uint64_t>(store, t2, P, s_ops, 0, cols, 1u);
    r2.extra_begin = (int16_t)(r2.extra_begin | 0x4000);
    t = t2; r = r2;
}

// the candidates k_path_fast listed, one block (one working lane, LDS scratch: see k_left_rescue) each
__global__ __launch_bounds__(64) void k_fix_boundary(const uint32_t *__restrict__ store, const uint32_t *__restrict__ list, const uint32_t *__restrict__ n_list_dev,
                                                     fsv_wtask *__restrict__ tasks, fsv_wres *__restrict__ res, fsv_wpath *__restrict__ paths, int k_cap,
                                                     uint32_t *__restrict__ n_fixed)
{
    __shared__ uint32_t s_ops[28][64];
    __shared__ uint64_t s_cols[(FSV_WINDOW + 2) * 3];
    if (threadIdx.x != 0) return;
    const uint32_t n_list = *n_list_dev;
    for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
        const uint32_t tid = list[li];
        fsv_wtask t = tasks[tid];
        fsv_wres r = res[tid];
        const int y0 = t.y_start;
        fix_boundary_dev(store, t, r, paths + tid, s_ops, s_cols, k_cap);
        if (t.y_start != y0) { tasks[tid] = t; res[tid] = r; if (n_fixed) atomicAdd(n_fixed, 1u); }
    }
}

// ------------------------------------------------------------------------------------------------ k_left_rescue
// recalcate_window_advance's left pass (Correct.cpp:2745-2905), for the overlaps k_rescue_accept set aside: a matched window whose
// left neighbour is unmatched gets its path first -- its real start on y -- and the unmatched windows to its left are tried again one
// after the other, each placed so that it ends right in front of the window to its right, with the doubled threshold and its path at
// once (the next one needs its start).  Then the overlap is accepted or not, as k_rescue_accept does: a window whose path was
// computed here counts with its distance after generate_cigar, as in hifiasm.  One block (one working lane) per listed overlap, a
// persistent grid over the list; oracle/asm.c:align_overlaps (left_rescue) statement for statement.
__global__ __launch_bounds__(64) void k_left_rescue(const uint32_t *__restrict__ store, fsv_ovl *__restrict__ ovl, const uint32_t *__restrict__ list,
                                                    const uint32_t *__restrict__ n_list_dev, fsv_wtask *__restrict__ tasks, fsv_wres *__restrict__ res,
                                                    fsv_wpath *__restrict__ paths, uint64_t *__restrict__ cols, uint4 *__restrict__ ovl_c, int k_cap, int accept_err_pm)
{
    // One lane of a block works, with the column scratch of its window in LDS: the walk back is a chain of dependent reads of that
    // scratch, column after column -- 1.3 ms for a single window from HBM, whatever the number of overlaps listed (a few thousand in the
    // first round, a dozen later); 40 us from LDS.  (cols: unused.)
    __shared__ uint32_t s_ops[28][64];
    __shared__ uint64_t s_cols[(FSV_WINDOW + 2) * 3];
    if (threadIdx.x != 0) return;
    const int lane64 = 0;
    const uint32_t n_list = *n_list_dev;
    uint64_t *slice = s_cols;
    (void)cols;
    for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
        const uint32_t p = list[li];
        fsv_ovl o = ovl[p];
        fsv_wtask *T = tasks + o.first_win;
        fsv_wres *R = res + o.first_win;
        fsv_wpath *PP = paths + o.first_win;
        long long post = 0;          // sum over the windows whose path was computed here of (distance after generate_cigar - K5's distance)
        auto window_path = [&](fsv_wtask &t, fsv_wres &r, fsv_wpath *P) {
            if (!path_gapfree(store, t, r, P, true)) path_general<uint64_t>(store, t, P, s_ops, lane64, slice, 1u);
            fix_boundary_dev(store, t, r, P, s_ops, slice, k_cap);
        };
        for (int j = 1; j < o.n_win; j++) {
            if (R[j].err < 0 || R[j - 1].err >= 0) continue;
            fsv_wtask tj = T[j];
            fsv_wres rj = R[j];
            const int raw_err_j = rj.err;
            window_path(tj, rj, PP + j);
            if (tj.y_start != T[j].y_start) { T[j] = tj; R[j] = rj; }      // (fix_boundary moved the window)
            const uint4 hj = *reinterpret_cast<const uint4 *>(PP + j);
            if ((hj.w & 0xffu) != 1u) { R[j].err = -1; continue; }       // (a path longer than a record holds: the window is unused, as in window_path)
            post += (int)(int16_t)(hj.z >> 16) - rj.err;
            (void)raw_err_j;
            int total_y_end = (int)hj.x - 1;
            for (int k2 = j - 1; k2 >= 0 && R[k2].err < 0; k2--) {
                fsv_wtask u = T[k2];
                u.k = (uint8_t)double_thr(u.k, u.x_len, k_cap);
                if (total_y_end <= 0) break;
                u.y_start = total_y_end - (int)u.x_len + 1;
                fsv_wres r;
                if (!bpm_window_geometry(u, r, k_cap)) break;
                if ((u.x_len + 2 * u.k - r.extra_begin - r.extra_end) + u.k < u.x_len) break;
                bpm_run(store, u, r, BpmNoSink());
                if (r.err < 0) break;
                window_path(u, r, PP + k2);
                const uint4 hk = *reinterpret_cast<const uint4 *>(PP + k2);
                if ((hk.w & 0xffu) != 1u) break;
                T[k2] = u; R[k2] = r;
                post += (int)(int16_t)(hk.z >> 16) - r.err;
                total_y_end = (int)hk.x - 1;
            }
        }
        int align = 0;
        long long tlen = 0, terr = post;
        for (int j = 0; j < o.n_win; j++) {
            const int e = R[j].err, xl = T[j].x_len;
            if (e >= 0) { align += xl; terr += e; } else terr += xl;
            tlen += xl;
        }
        o.align_len = align; o.err_sum = (int32_t)terr;
        o.is_match = ((long long)(o.x_e - o.x_s + 1) * 9 <= (long long)align * 10 && terr * 1000 <= tlen * accept_err_pm) ? 1 : 0;
        ovl[p] = o;
        ovl_c[p] = make_uint4((uint32_t)o.x_s, (uint32_t)o.first_win, (uint32_t)o.n_win | (o.is_match ? 0x80000000u : 0u), 0u);
    }
}

// ---- K6 for first-pass windows: k <= 15, distance 4 .. FSV_SB_MAXERR (3 and below: k_path_fr further down) ----------------
// What the walk back (Levenshtein_distance.h:757-888) asks of a DP cell is which way it leaves it -- 0 diagonal over a match,
// 1 diagonal over a mismatch, 2 up, 3 left; ties: diagonal, then up, then left -- and that is known while the column is
// computed: a cell whose D0 bit is clear is a mismatch (the diagonal is one cheaper than the cell, nothing beats it);
// otherwise "up" is cheaper exactly when the column's new VP has the bit of the row below set, and "left" when HP has the
// cell's bit (the top band row has no left neighbour).  And the walk never strays further than `err` rows from the end row:
// every up / left step spends one of the err errors it has left.  K5 already gave (end site, err) for the window, so the
// forward pass keeps two bits for each of the 2 x 7 + 1 rows around the end row: ONE 32-bit word per column (bits 0-15 the
// low code bit of rows row0-7 .. row0+8, bits 16-31 the high one) instead of three band-wide words.  Columns go to the scratch
// four at a time ([block][column quad][lane] as uint4: 1 KB per wave store); the walk reads them back a quad ahead of where it
// stands, so no step waits on memory -- round 1's walk was a chain of ~375 dependent loads per window (62 % of its wave cycles
// parked in s_waitcnt, profiles/r01_i_pmc_sq_summary.txt).  Scratch traffic: 1.5 KB per window, written once, read once.
struct SubbandSink {
    uint4 *slot;             // this lane's uint4 of quad 0; quad q sits 64 x q further
    uint32_t sr, sl, lmask;  // band word -> sub-band: (w >> sr) << sl; rows that may step left
    uint32_t a0, a1, a2, a3;
    __device__ __forceinline__ void operator()(int blk, int j, uint32_t d0, uint32_t hp, uint32_t vp, uint32_t)
    {
        const uint32_t u = vp << 1, l = hp & lmask;
        const uint32_t w1 = d0 & (u | l), w0 = ~d0 | (l & ~u);
        const uint32_t word = (((w0 >> sr) << sl) & 0xffffu) | (((w1 >> sr) << sl) << 16);
        if ((j & 3) == 0) a0 = word; else if ((j & 3) == 1) a1 = word; else if ((j & 3) == 2) a2 = word; else a3 = word;
        if ((j & 3) == 3) slot[(size_t)((blk + j) >> 2) * 64] = make_uint4(a0, a1, a2, a3);
    }
    __device__ __forceinline__ void flush(int n) { if (n & 3) slot[(size_t)(n >> 2) * 64] = make_uint4(a0, a1, a2, a3); }
};

__device__ __forceinline__ uint32_t quad_elem(const uint4 &q, int e) { return e == 0 ? q.x : e == 1 ? q.y : e == 2 ? q.z : q.w; }

// STAMP: diagnostic build only (FSV_K6_STAMPS=1): shader-clock cycles of the three phases summed per wave into `stamps`
template <bool STAMP>
__global__ __launch_bounds__(64) void k_path_sb(const uint32_t *__restrict__ store, const fsv_wtask *__restrict__ tasks, const fsv_wres *__restrict__ res,
                                                const uint32_t *__restrict__ dp_list, const uint32_t *__restrict__ n_dev,
                                                fsv_wpath *__restrict__ paths, uint4 *__restrict__ cols, unsigned long long *__restrict__ stamps)
{
    unsigned long long t_fwd = 0, t_walk = 0, t_fin = 0, t0 = 0, t1 = 0, t2 = 0;
    __shared__ uint32_t s_ops[28][64];     // per lane: the path being built, 2 bits per op, stored end-to-start (448 ops)
    const int lane64 = threadIdx.x;
    const uint32_t n_list = *n_dev;
    uint4 *slot = cols + (size_t)blockIdx.x * FSV_SB_QUADS * 64 + lane64;   // persistent blocks: the slice is reused for every task
    for (uint32_t li = blockIdx.x * 64 + threadIdx.x; li < n_list; li += gridDim.x * 64) {
        const uint32_t tid = dp_list[li];
        const fsv_wtask t = tasks[tid];
        const fsv_wres r0 = res[tid];
        fsv_wpath *P = paths + tid;
        const int n = t.x_len, k = t.k, band = 2 * k + 1;
        const int end = r0.end_site, err = r0.err;
        constexpr int ME = FSV_SB_MAXERR, QSH = 2;
        const int row0 = band - (n + 2 * k - end), lo = row0 - ME;
        SubbandSink sink;
        sink.slot = slot; sink.sr = (uint32_t)max(lo, 0); sink.sl = (uint32_t)max(-lo, 0);
        sink.lmask = band == 1 ? 1u : (1u << (band - 1)) - 1u;
        sink.a0 = sink.a1 = sink.a2 = sink.a3 = 0;
        fsv_wres r;
        if (STAMP) t0 = __builtin_amdgcn_s_memtime();
        bpm_run32(store, t, r, sink);
        if (STAMP) t1 = __builtin_amdgcn_s_memtime();
        if (r.err != err || r.end_site != end) { P->state = 0; continue; }   // cannot happen: the same DP as K5
        for (int i = 0; i < 28; i++) s_ops[i][lane64] = 0;
        int cur = err, ci = n - 1, plen = 0, start = end, rel = ME, dir = 0;
        uint32_t acc = 0;
        // The walk, a quad of columns per phase: every lane walks until it leaves its current quad (four column steps plus its
        // "up" steps), then all lanes move one quad down together.  Six quads rotate through registers and the one just left
        // is refilled with the quad six below, so a quad is requested five phases before it is walked and no lane ever waits
        // for a load another lane has just issued (with a per-lane "switch when I cross" every crossing waited out the full
        // memory latency of the neighbour's request: 1 500 cycles per step, FSV_K6_STAMPS).
        int qi = ci >> QSH;
        auto quad = [&](int q) { return q >= 0 ? slot[(size_t)q * 64] : make_uint4(0, 0, 0, 0); };
        uint4 qa = quad(qi), qb = quad(qi - 1), qc = quad(qi - 2), qd = quad(qi - 3), qe = quad(qi - 4), qf = quad(qi - 5);
        auto phase = [&](const uint4 &q4) {
            while (cur != 0 && ci >= 0 && (ci >> QSH) == qi) {
                // A match step keeps the band row (`rel`) and moves one column left, so a run of matches is a run of zero codes at ONE
                // bit position of consecutive columns' words: the columns of this quad whose code at `rel` is not 0 are found with a few
                // shifts, and the matches in front of the first of them are taken in one step (round 2 walked them one by one, ~40
                // instructions each: half of K6's time by the cycle stamps, 375 steps for the 1-3 deviations of a HiFi window).
                uint32_t nz;
                nz = (((q4.x >> rel) | (q4.x >> (rel + 16))) & 1u) | ((((q4.y >> rel) | (q4.y >> (rel + 16))) & 1u) << 1) |
                     ((((q4.z >> rel) | (q4.z >> (rel + 16))) & 1u) << 2) | ((((q4.w >> rel) | (q4.w >> (rel + 16))) & 1u) << 3);
                const int cl = ci & 3;
                const uint32_t m = nz & ((2u << cl) - 1u);          // deviating columns at or below this one
                const int steps = m ? cl - (31 - __clz((int)m)) : cl + 1;
                if (steps) {
                    const int np = plen + steps;
                    if ((np >> 4) != (plen >> 4)) { s_ops[plen >> 4][lane64] = acc; acc = 0; }   // the word fills up with matches
                    plen = np; start -= steps; ci -= steps; dir = 0;
                    if (!m) continue;                                   // the rest of the quad matched
                }
                const uint32_t w = quad_elem(q4, ci & 3);
                const uint32_t code = ((w >> rel) & 1u) | (((w >> (16 + rel)) & 1u) << 1);
                acc |= code << ((plen & 15) << 1);
                if ((plen & 15) == 15) { s_ops[plen >> 4][lane64] = acc; acc = 0; }
                plen++;
                cur -= (int)(code != 0u);
                start -= (int)(code != 3u);
                rel += (int)(code == 3u) - (int)(code == 2u);
                ci -= (int)(code != 2u);      // "up" stays in its column
                dir = (int)code;
            }
        };
        // (six quads in rotation since round 3: a quad is requested five phases before it is walked.  With three -- two phases, ~800
        // cycles of walking -- every phase still waited out most of a memory round trip: the stamps showed half of K6's time in the walk)
        while (__any(cur != 0 && ci >= 0)) {
            phase(qa); qi--; qa = quad(qi - 5);
            phase(qb); qi--; qb = quad(qi - 5);
            phase(qc); qi--; qc = quad(qi - 5);
            phase(qd); qi--; qd = quad(qi - 5);
            phase(qe); qi--; qe = quad(qi - 5);
            phase(qf); qi--; qf = quad(qi - 5);
        }
        if (plen & 15) s_ops[plen >> 4][lane64] = acc;
        if (STAMP) t2 = __builtin_amdgcn_s_memtime();
        path_finish(store, t, P, s_ops, lane64, ci + 1, dir, plen, start, end, err);
        if (STAMP) { const unsigned long long t3 = __builtin_amdgcn_s_memtime(); t_fwd += t1 - t0; t_walk += t2 - t1; t_fin += t3 - t2; }
    }
    if (STAMP && lane64 == 0) { atomicAdd(&stamps[0], t_fwd); atomicAdd(&stamps[1], t_walk); atomicAdd(&stamps[2], t_fin); atomicAdd(&stamps[3], 1ull); }
}

// ---- K6 without the matrix: distance <= FSV_FR_MAXERR ------------------------------------------------------------------------
// The walk back (Levenshtein_distance.h:757-888) asks three things of the cell (column c, band row r) it stands on, whose
// distance `cur` it knows: is the diagonal neighbour (c-1, r) one cheaper (a mismatch), else is the upper one (c, r-1), else the
// left one (c-1, r+1); none of them: a match, one column down the same row.  A band row is a diagonal of the alignment
// matrix and the distance never falls along a diagonal, so "cell (c, r) is within s errors" is c <= F[s][r], the furthest
// column row r reaches with s errors -- Landau-Vishkin's table, started from the free start of the DP (every band row at
// column -1 with distance 0) and clipped to the band:
//     F[0][r] = last column of the run of matches from column 0 on row r
//     F[s][r] = the run of matches behind max(F[s-1][r] + 1, F[s-1][r-1], F[s-1][r+1] + 1)      (mismatch, up, left)
// The walk starts on the end row with the window's distance e (K5 gave both) and spends an error with every move off a match,
// so the cell it stands on with `cur` left is at most e - cur rows from the end row and the three neighbours it tests sit on
// level cur - 1 at most e - cur + 1 rows away: a triangle of (2e+1) + (2e-1) + .. + 3 table entries (15 for e = 3), each a
// word-wise comparison of packed bases, instead of 375 columns of the recurrence and their 750-byte scratch.  Between two
// errors the walk is one subtraction: it matches down its row to the first column where a neighbour opens.
// tests/test_gpu_k6.py holds this kernel to the oracle's matrix walk.
// first column in [cs, cs + 64) (none at or past n) whose x base differs from the y base at strand position ypos + column, as an offset from cs
__device__ __forceinline__ int diag_mismatch16(uint32_t xb, uint32_t yb, uint32_t yvalid)
{
    uint32_t d = xb ^ yb;
    d = (d | (d >> 1)) & 0x55555555u;
    uint32_t inval = ~yvalid & 0xffffu;     // columns outside read y never match
    inval = (inval | (inval << 8)) & 0x00ff00ffu; inval = (inval | (inval << 4)) & 0x0f0f0f0fu;
    inval = (inval | (inval << 2)) & 0x33333333u; inval = (inval | (inval << 1)) & 0x55555555u;
    d |= inval;
    return d ? (__ffs((int)d) - 1) >> 1 : 16;
}
__device__ __forceinline__ int diag_probe64(const uint32_t *__restrict__ store, const fsv_wtask &t, int ypos, int cs, int n)
{
    uint32_t xb4[4], yb4[4], yv4[4];
    fetch64_x(store, t.x_word, t.x_start + cs, xb4);
    fetch64(store, t.y_word, t.y_len, t.y_rev, ypos + cs, yb4, yv4);
    int m = 64;
#pragma unroll
    for (int q = 3; q >= 0; q--) { const int f = diag_mismatch16(xb4[q], yb4[q], yv4[q]); if (f < 16) m = q * 16 + f; }
    return min(m, n - cs);
}

// E = the windows' distance (one list per distance: a wave's lanes then have the same number of table entries to fill)
// STAMP: diagnostic build only (FSV_K6_STAMPS=1): shader-clock cycles of table / walk / finish summed per wave into `stamps`
template <int E, bool STAMP = false>
__global__ __launch_bounds__(64) void k_path_fr(const uint32_t *__restrict__ store, const fsv_wtask *__restrict__ tasks, const fsv_wres *__restrict__ res,
                                                const uint32_t *__restrict__ dp_list, const uint32_t *__restrict__ n_dev, fsv_wpath *__restrict__ paths,
                                                unsigned long long *__restrict__ stamps = nullptr)
{
    unsigned long long t_tab = 0, t_walk = 0, t_fin = 0, t0 = 0, t1 = 0, t2 = 0;
    __shared__ uint32_t s_ops[28][64];     // per lane: the path, 2 bits per op, end-to-start (path_finish works on it)
    constexpr int NJ = 2 * E + 1;
    const int lane64 = threadIdx.x;
    const uint32_t n_list = *n_dev;
    for (uint32_t li = blockIdx.x * 64 + threadIdx.x; li < n_list; li += gridDim.x * 64) {
        if (STAMP) t0 = __builtin_amdgcn_s_memtime();
        const uint32_t tid = dp_list[li];
        const fsv_wtask t = tasks[tid];
        const fsv_wres r0 = res[tid];
        fsv_wpath *P = paths + tid;
        const int n = t.x_len, k = t.k, band = 2 * k + 1;
        const int end = r0.end_site;
        const int row0 = band - (n + 2 * k - end), win0 = t.y_start - k;
        // Q[s][j + E] = F[s][row0 + j] + 1: the first column of the row that is NOT within s errors; -1 = no such row (outside the
        // band or the triangle): F = -2 is below every column the walk can ask about
        int Q[E][NJ];
        // the rows in `run` lengthen their runs of matches, 64 columns a trip, every lane working on its lowest running row: the
        // wave makes as many trips as its busiest lane has chunks to compare (row after row it made the sum of the rows' longest)
        auto extend = [&](int (&q)[NJ], uint32_t run) {
            while (__any(run != 0u)) {
                if (run) {
                    const int jj = __ffs((int)run) - 1;
                    int c = 0;
#pragma unroll
                    for (int i = 0; i < NJ; i++) c = i == jj ? q[i] : c;
                    const int m = diag_probe64(store, t, win0 + row0 + jj - E, c, n);
                    c += m;
#pragma unroll
                    for (int i = 0; i < NJ; i++) q[i] = i == jj ? c : q[i];
                    if (m < 64 || c >= n) run &= run - 1u;
                }
            }
        };
        {
            // level 0, first 64 columns: every row starts at column 0 and the rows' y bases overlap -- one fetch for all of them
            uint32_t xb4[4], yb[5], yv[5];
            fetch64_x(store, t.x_word, t.x_start, xb4);
            {
                uint32_t b4[4], v4[4];
                fetch64(store, t.y_word, t.y_len, t.y_rev, win0 + row0 - E, b4, v4);
                const Bases16 b5 = fetch16(store, t.y_word, t.y_len, t.y_rev, win0 + row0 - E + 64);
#pragma unroll
                for (int q = 0; q < 4; q++) { yb[q] = b4[q]; yv[q] = v4[q] & 0xffffu; }
                yb[4] = b5.bits; yv[4] = b5.valid & 0xffffu;
            }
            uint32_t run = 0;
#pragma unroll
            for (int jj = 0; jj < NJ; jj++) {
                const int row = row0 + jj - E;
                int m = -1;
                if (row >= 0 && row < band) {
                    m = 64;
#pragma unroll
                    for (int q = 3; q >= 0; q--) {
                        const int f = diag_mismatch16(xb4[q], __builtin_amdgcn_alignbit(yb[q + 1], yb[q], 2 * jj), ((yv[q] | yv[q + 1] << 16) >> jj) & 0xffffu);
                        if (f < 16) m = q * 16 + f;
                    }
                    m = min(m, n);
                    if (m == 64 && n > 64) run |= 1u << jj;
                }
                Q[0][jj] = m;
            }
            extend(Q[0], run);
        }
#pragma unroll
        for (int s = 1; s < E; s++) {
            uint32_t run = 0;
#pragma unroll
            for (int jj = 0; jj < NJ; jj++) {
                const int row = row0 + jj - E;
                int c = -1;
                if (abs(jj - E) <= E - s && row >= 0 && row < band) {
                    // the furthest cell of the row within s errors before its matches: behind a mismatch on the row, an "up" move
                    // from the row below (same column), a "left" move from the row above (next column)
                    const int up = jj > 0 ? Q[s - 1][jj - 1] - 1 : -2, left = jj + 1 < NJ ? Q[s - 1][jj + 1] : -2;
                    c = min(n - 1, max(Q[s - 1][jj], max(up, left))) + 1;
                    if (c < n) run |= 1u << jj;
                }
                Q[s][jj] = c;
            }
            extend(Q[s], run);
        }
        if (STAMP) t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 28; i++) s_ops[i][lane64] = 0;
        int ci = n - 1, plen = 0, start = end, j = E, dir = 0;
#pragma unroll
        for (int s = E - 1; s >= 0; s--) {      // the walk has s + 1 errors left: its neighbours are judged on level s
            int fa = -1, fb = -2, fc = -1;      // first columns (from the end) where the mismatch / up / left move is open
#pragma unroll
            for (int i = 0; i < NJ; i++) { fa = i == j ? Q[s][i] : fa; fb = i == j - 1 ? Q[s][i] - 1 : fb; fc = i == j + 1 ? Q[s][i] : fc; }
            const int cstop = min(ci, max(fa, max(fb, fc)));    // fa >= 0: column 0 is a mismatch at the latest
            const int steps = ci - cstop;
            plen += steps; start -= steps; ci = cstop;
            const uint32_t code = ci <= fa ? 1u : ci <= fb ? 2u : 3u;
            s_ops[plen >> 4][lane64] |= code << ((plen & 15) << 1);
            plen++;
            start -= (int)(code != 3u);
            j += (int)(code == 3u) - (int)(code == 2u);
            ci -= (int)(code != 2u);
            dir = (int)code;
        }
        if (STAMP) t2 = __builtin_amdgcn_s_memtime();
        path_finish(store, t, P, s_ops, lane64, ci + 1, dir, plen, start, end, E);
        if (STAMP) { const unsigned long long t3 = __builtin_amdgcn_s_memtime(); t_tab += t1 - t0; t_walk += t2 - t1; t_fin += t3 - t2; }
    }
    if (STAMP && lane64 == 0) { atomicAdd(&stamps[0], t_tab); atomicAdd(&stamps[1], t_walk); atomicAdd(&stamps[2], t_fin); atomicAdd(&stamps[3], 1ull); }
}

// ---- K6 for wide bands (k > 31, up to 95): the ONT profile ---------------------------------------------------------------------
// The walk codes of k_path_sb for every band row: two planes x six 32-bit limbs per column (48 B) in a per-lane slice of an HBM
// scratch ([block][column][12][lane]).  The distance of such a window is of the order of its band (tens of errors), so there is
// no narrow sub-band to keep, and the walk reads the two words of its row's limb at every step.  First version: correct, not
// tuned (the forward pass is ~130 lane-ops per column, the walk a dependent load per step).
struct WideCodeSink {
    uint32_t *cols; uint32_t lane; int band;
    __device__ __forceinline__ void operator()(int col, const uint32_t *d0, const uint32_t *hp, const uint32_t *vp)
    {
        uint32_t *c = cols + (size_t)col * 2 * FSV_WL * 64 + lane;
#pragma unroll
        for (int l = 0; l < FSV_WL; l++) {
            const uint32_t u = vp[l] << 1 | (l ? vp[l - 1] >> 31 : 0u);
            // the top band row has no left neighbour
            const int tb = band - 1 - 32 * l;
            const uint32_t lm = band == 1 ? 1u : (tb >= 32 ? 0xffffffffu : (tb <= 0 ? 0u : ((1u << tb) - 1u)));
            const uint32_t lf = hp[l] & lm;
            c[(size_t)l * 64] = ~d0[l] | (lf & ~u);              // low code bit
            c[(size_t)(FSV_WL + l) * 64] = d0[l] & (u | lf);     // high code bit
        }
    }
};

__global__ __launch_bounds__(64) void k_path_wide(const uint32_t *__restrict__ store, const fsv_wtask *__restrict__ tasks, const fsv_wres *__restrict__ res,
                                                  const uint32_t *__restrict__ dp_list, const uint32_t *__restrict__ n_dev,
                                                  fsv_wpath *__restrict__ paths, uint32_t *__restrict__ cols, int k_cap)
{
    __shared__ uint32_t s_ops[28][64];
    const int lane64 = threadIdx.x;
    const uint32_t n_list = *n_dev;
    uint32_t *slot = cols + (size_t)blockIdx.x * FSV_WINDOW * 2 * FSV_WL * 64;
    for (uint32_t li = blockIdx.x * 64 + threadIdx.x; li < n_list; li += gridDim.x * 64) {
        const uint32_t tid = dp_list[li];
        const fsv_wtask t = tasks[tid];
        const fsv_wres r0 = res[tid];
        fsv_wpath *P = paths + tid;
        const int n = t.x_len, k = t.k, band = 2 * k + 1;
        const int end = r0.end_site, err = r0.err;
        WideCodeSink sink{slot, (uint32_t)lane64, band};
        fsv_wres r;
        bpm_run_wide(store, t, r, sink, k_cap);
        if (r.err != err || r.end_site != end) { P->state = 0; continue; }   // cannot happen: the same DP as K5
        for (int i = 0; i < 28; i++) s_ops[i][lane64] = 0;
        int cur = err, ci = n - 1, plen = 0, start = end, row = band - (n + 2 * k - end), dir = 0;
        uint32_t acc = 0;
        bool fits = true;
        // The walk, column by column for the whole wavefront: the two code words a lane needs at a column (those of its row's limb)
        // are requested FOUR columns ahead, at a point every lane passes together, and held in four register pairs that an unrolled
        // loop uses in turn -- round 2's walk was a dependent pair of loads per step (~400 steps of full memory latency per window:
        // 314 ms per ONT step).  A lane whose row has moved to another limb by the time it reaches a column (a limb is 32 rows; the
        // row drifts by one per gap) loads that column's words directly.
        auto ld = [&](int c, int lm, uint32_t &lo, uint32_t &hi) {
            if (c >= 0) { const uint32_t *p = slot + (size_t)c * 2 * FSV_WL * 64 + lane64; lo = p[(size_t)lm * 64]; hi = p[(size_t)(FSV_WL + lm) * 64]; }
        };
        auto column = [&](int c, uint32_t &lo, uint32_t &hi, int &lmx) {
            if (c >= 0 && ci == c && cur != 0 && fits) {
                while (true) {
                    const int lm = row >> 5, bt = row & 31;
                    if (lm != lmx) { ld(c, lm, lo, hi); lmx = lm; }
                    const uint32_t code = ((lo >> bt) & 1u) | (((hi >> bt) & 1u) << 1);
                    if (plen >= 28 * 16 - 1) { fits = false; break; }        // the path buffer holds 448 ops; such a path is dropped below anyway
                    acc |= code << ((plen & 15) << 1);
                    if ((plen & 15) == 15) { s_ops[plen >> 4][lane64] = acc; acc = 0; }
                    plen++;
                    cur -= (int)(code != 0u);
                    start -= (int)(code != 3u);
                    row += (int)(code == 3u) - (int)(code == 2u);
                    dir = (int)code;
                    if (code != 2u) { ci--; break; }     // "up" stays in its column
                    if (cur == 0) break;
                }
            }
            lmx = row >> 5;
            ld(c - 4, lmx, lo, hi);          // (every lane, walking or not: the request is issued where the whole wave passes)
        };
        const int c0 = FSV_WINDOW - 1;       // every window has at most FSV_WINDOW columns; a shorter one idles until the loop reaches its last column
        uint32_t lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0, lo2 = 0, hi2 = 0, lo3 = 0, hi3 = 0;
        int lm0 = row >> 5, lm1 = lm0, lm2 = lm0, lm3 = lm0;
        ld(c0, lm0, lo0, hi0); ld(c0 - 1, lm1, lo1, hi1); ld(c0 - 2, lm2, lo2, hi2); ld(c0 - 3, lm3, lo3, hi3);
        for (int c = c0; c >= 0; c -= 4) {
            if (!__any(cur != 0 && ci >= 0 && fits)) break;
            column(c, lo0, hi0, lm0);
            column(c - 1, lo1, hi1, lm1);
            column(c - 2, lo2, hi2, lm2);
            column(c - 3, lo3, hi3, lm3);
        }
        if (plen & 15) s_ops[plen >> 4][lane64] = acc;
        if (!fits) { P->state = 0; continue; }
        path_finish(store, t, P, s_ops, lane64, ci + 1, dir, plen, start, end, err);
    }
}

// ------------------------------------------------------------------------------------------------ k_consensus
// One wavefront per (read, 375-bp grid window).  Lanes walk the window paths of the accepted overlaps
// and vote into LDS histograms (per column: A C G T deleted arrived after-insertion); inserted strings are
// kept as (column, key) events.  Then lanes take columns, decide, and the corrected window is written out.
struct ConsArgs {
    const uint32_t *store;
    const uint32_t *word_off;
    const int32_t *read_len;
    const uint32_t *read_set;    // read -> set
    const uint32_t *set_start;
    const uint32_t *pair_base;
    const uint32_t *gwin_off;    // n_reads + 1: first grid window of every read
    const uint32_t *gwin_read;   // n_gwin: read of every grid window
    const uint4 *ovl_c;          // per ordered pair: {x_s, first window task, n_win | accepted << 31, -}  (k_rescue_accept)
    const uint4 *gwin_tab;       // per grid window: {read, first pair slot of the read, overlaps of the read, window index}  (k_gwin_tab)
    const fsv_wtask *tasks;
    const fsv_wpath *paths;
    uint8_t *cwin;               // FSV_CW_STRIDE bytes per grid window (2-bit codes, one per byte)
    uint16_t *cwin_len;
    uint32_t *warn;
    uint32_t *changed;           // per read: set when the consensus of some window differs from the read (nullptr: not tracked)
    uint32_t n_reads;
    const uint32_t *read_dirty;  // per read: some accepted overlap deviates from it somewhere (k_read_dirty); nullptr: not known
    uint8_t *cov3;               // per grid window: at least three overlaps voted (the window went through window_consensus); nullptr: not kept
    int junction_vote;           // 1: bases skipped between two windows of an overlap are voted as an insertion (the stand-in for the second pass)
    int ins_dag;                 // 1: inserted strings that disagree go through hifiasm's DAG (lane 0); 0: the most frequent string (ONT profile)
};

__device__ __forceinline__ bool vote_wins(int cnt, int total, bool homo)
{
    if (cnt * 5 >= total * 3) return true;
    return homo && cnt * 1000 >= total * 515;
}

// One wavefront per read, one lane per overlap: does some accepted overlap deviate from the read anywhere -- a window at distance
// > 0, or y bases skipped between two consecutive windows (what k_consensus votes as a junction insertion)?  A read no overlap
// deviates from keeps every window as it is; from the second round on that is most reads, and their windows skip the tally.
__global__ __launch_bounds__(64) void k_read_dirty(const uint4 *__restrict__ ovl_c, const fsv_wpath *__restrict__ paths, const uint32_t *__restrict__ read_set,
                                                   const uint32_t *__restrict__ set_start, const uint32_t *__restrict__ pair_base, uint32_t n_reads,
                                                   uint32_t *__restrict__ read_dirty)
{
    const uint32_t r = blockIdx.x;
    if (r >= n_reads) return;
    const uint32_t s = read_set[r], r0 = set_start[s], ns = set_start[s + 1] - r0;
    const uint32_t pbase = pair_base[s] + (r - r0) * (ns - 1), n_ovl = ns - 1;
    bool dirty = false;
    for (uint32_t oi = threadIdx.x; oi < n_ovl; oi += 64) {
        const uint4 oc = ovl_c[pbase + oi];
        if (!(oc.z >> 31)) continue;
        const int n_win = (int)(oc.z & 0x7fffffffu);
        int prev_end = 0; bool prev_ok = false;
        for (int j = 0; j < n_win && !dirty; j++) {
            const uint4 h0 = *reinterpret_cast<const uint4 *>(paths + (oc.y + (uint32_t)j));
            const bool ok = (h0.w & 0xffu) == 1u;
            if (ok) {
                if ((int16_t)(h0.z >> 16) != 0) dirty = true;
                if (prev_ok && (int)h0.x - prev_end - 1 != 0) dirty = true;     // bases of y skipped, or used twice
                prev_end = (int)h0.y;
            }
            prev_ok = ok;
        }
    }
    const bool any = __ballot(dirty) != 0ull;
    if (threadIdx.x == 0) read_dirty[r] = any ? 1u : 0u;
}

// per grid window, once per round: everything k_consensus would otherwise look up through three levels of tables
__global__ void k_gwin_tab(const uint32_t *__restrict__ gwin_read, const uint32_t *__restrict__ gwin_off, const uint32_t *__restrict__ read_set,
                           const uint32_t *__restrict__ set_start, const uint32_t *__restrict__ pair_base, uint32_t n_gwin, uint4 *__restrict__ tab)
{
    const uint32_t gw = blockIdx.x * blockDim.x + threadIdx.x;
    if (gw >= n_gwin) return;
    const uint32_t r = gwin_read[gw], s = read_set[r], r0 = set_start[s], ns = set_start[s + 1] - r0;
    tab[gw] = make_uint4(r, pair_base[s] + (r - r0) * (ns - 1), ns - 1, gw - gwin_off[r]);
}

// EVC: insertion events a window can hold (HiFi at 30x: ~8 -> FSV_EV_CAP; ONT-profile reads: hundreds -> FSV_EV_CAP_WIDE)
// MODE 0: every window.  MODE 1: every window, and a window with a column that split_sub_list would keep as a site of the
// haplotype partition is marked (site_cnt[gw] = FSV_SITE_MARK) -- the consensus written here is then provisional: k_snp_sites and
// k_hap_partition run next, and the windows of a read that lost overlaps to the partition are redone.  MODE 2: that redo.
#define FSV_SITE_MARK 0xffffffffu
// ---- what is inserted in front of a column: hifiasm's DAG of the inserted strings ---------------------------------------------
// build_DAGCon / Merge_DAGCon / generate_best_seq_from_nodes (Correct.cpp:3219-3951), as oracle/asm.c:dagcon_insertion restates
// them: one chain S -> b1 -> ... -> E per distinct string (here in ascending key order) weighted by its count, nodes gone through
// in topological order merging per base the in-nodes with one out-edge and the out-nodes with one in-edge, node weight = sum of the
// out-edges (E: in-edges), greedy walk forward from S's heaviest out-node or backward from E's heaviest in-node.  One lane runs it
// on a scratch in LDS; beyond the bounds (FSV_DG_*) the caller inserts the most frequent string instead, as the oracle does.
#define FSV_DG_N 64
#define FSV_DG_E 128
#define FSV_DG_A 8
#define FSV_DG_D 8
#define FSV_DG_K 64
struct DagLds {
    uint32_t keys[FSV_DG_K];
    uint16_t e_w[FSV_DG_E];
    uint8_t e_from[FSV_DG_E], e_to[FSV_DG_E], e_alive[FSV_DG_E], e_vis[FSV_DG_E];
    uint8_t base[FSV_DG_N], alive[FSV_DG_N], out_n[FSV_DG_N], in_n[FSV_DG_N];
    uint8_t out_e[FSV_DG_N][FSV_DG_A], in_e[FSV_DG_N][FSV_DG_A];
    uint8_t queue[4 * FSV_DG_E];
    uint8_t stk_node[16], stk_bi[16], stk_cons[16];
    int n_node, n_edge, ok;
};

__device__ __forceinline__ int dg_node(DagLds &D, uint8_t b)
{
    const int id = D.n_node;
    if (id >= FSV_DG_N) { D.ok = 0; return FSV_DG_N - 1; }
    D.base[id] = b; D.alive[id] = 1; D.out_n[id] = 0; D.in_n[id] = 0; D.n_node = id + 1;
    return id;
}
__device__ __forceinline__ void dg_edge(DagLds &D, int u, int v, int w, int vis)
{
    const int e = D.n_edge;
    if (e >= FSV_DG_E || D.out_n[u] >= FSV_DG_A || D.in_n[v] >= FSV_DG_A) { D.ok = 0; return; }
    D.e_from[e] = (uint8_t)u; D.e_to[e] = (uint8_t)v; D.e_w[e] = (uint16_t)w; D.e_alive[e] = 1; D.e_vis[e] = (uint8_t)vis; D.n_edge = e + 1;
    D.out_e[u][D.out_n[u]++] = (uint8_t)e; D.in_e[v][D.in_n[v]++] = (uint8_t)e;
}
__device__ __forceinline__ int dg_find(const DagLds &D, int u, int v)
{
    for (int i = 0; i < D.in_n[v]; i++) { const int e = D.in_e[v][i]; if (D.e_alive[e] && D.e_from[e] == u) return e; }
    return -1;
}
__device__ __forceinline__ int dg_outdeg(const DagLds &D, int u) { int c = 0; for (int i = 0; i < D.out_n[u]; i++) c += D.e_alive[D.out_e[u][i]]; return c; }
__device__ __forceinline__ int dg_indeg(const DagLds &D, int u) { int c = 0; for (int i = 0; i < D.in_n[u]; i++) c += D.e_alive[D.in_e[u][i]]; return c; }
__device__ __forceinline__ void dg_delete(DagLds &D, int x)
{
    D.alive[x] = 0; D.base[x] = 'D';
    for (int i = 0; i < D.out_n[x]; i++) D.e_alive[D.out_e[x][i]] = 0;
    for (int i = 0; i < D.in_n[x]; i++) D.e_alive[D.in_e[x][i]] = 0;
    D.out_n[x] = 0; D.in_n[x] = 0;
}
// Merge_Out_Nodes (OUT) / Merge_In_Nodes (!OUT) with the recursion of the reference unrolled onto a small stack: a frame is
// (node, next base); after the merges for one base the merged node is entered before the next base is looked at
template <bool OUT>
__device__ __forceinline__ void dg_merge(DagLds &D, int start)
{
    int sp = 0;
    D.stk_node[0] = (uint8_t)start; D.stk_bi[0] = 0;
    if (!D.alive[start] || (OUT ? dg_outdeg(D, start) : dg_indeg(D, start)) == 0) return;
    while (sp >= 0 && D.ok) {
        const int cur = D.stk_node[sp], bi = D.stk_bi[sp];
        if (bi >= 4) { sp--; continue; }
        D.stk_bi[sp] = (uint8_t)(bi + 1);
        const uint8_t want = (uint8_t)("ACGT"[bi]);
        int flag = 0, weight = 0, cons = -1;
        const int nl = OUT ? D.out_n[cur] : D.in_n[cur];
        for (int i = 0; i < nl; i++) {
            const int e = OUT ? D.out_e[cur][i] : D.in_e[cur][i];
            if (!D.e_alive[e]) continue;
            const int g = OUT ? D.e_to[e] : D.e_from[e];
            if (D.base[g] != want || (OUT ? dg_indeg(D, g) : dg_outdeg(D, g)) != 1) continue;
            if (flag == 0) { flag = 1; cons = g; D.e_vis[e] = 1; weight = D.e_w[e]; }
            else {
                flag++;
                weight += D.e_w[e];
                const int ng = OUT ? D.out_n[g] : D.in_n[g];
                for (int j = 0; j < ng; j++) {
                    const int e2 = OUT ? D.out_e[g][j] : D.in_e[g][j];
                    if (!D.e_alive[e2]) continue;
                    const int o = OUT ? D.e_to[e2] : D.e_from[e2];
                    const int e3 = OUT ? dg_find(D, cons, o) : dg_find(D, o, cons);
                    if (e3 >= 0) { D.e_vis[e3] = 1; D.e_w[e3] = (uint16_t)(D.e_w[e3] + D.e_w[e2]); }
                    else if (OUT) dg_edge(D, cons, o, D.e_w[e2], 1);
                    else dg_edge(D, o, cons, D.e_w[e2], 1);
                }
                dg_delete(D, g);
            }
        }
        if (flag > 1) { const int e = OUT ? dg_find(D, cur, cons) : dg_find(D, cons, cur); if (e >= 0) D.e_w[e] = (uint16_t)weight; }
        if (flag > 0 && D.alive[cons] && (OUT ? dg_outdeg(D, cons) : dg_indeg(D, cons)) != 0) {
            if (sp + 1 >= 16) { D.ok = 0; return; }
            sp++;
            D.stk_node[sp] = (uint8_t)cons; D.stk_bi[sp] = 0;
        }
    }
}
__device__ __forceinline__ int dg_weight(const DagLds &D, int u, bool in)
{
    int w = 0;
    if (in) { for (int i = 0; i < D.in_n[u]; i++) if (D.e_alive[D.in_e[u][i]]) w += D.e_w[D.in_e[u][i]]; }
    else for (int i = 0; i < D.out_n[u]; i++) if (D.e_alive[D.out_e[u][i]]) w += D.e_w[D.out_e[u][i]];
    return w;
}
// D.keys[0 .. nk): the column's inserted strings (len << 24 | 2-bit bases).  Returns max_insertion_count (-1: beyond the bounds)
__device__ __forceinline__ int dag_insertion(DagLds &D, int nk, uint32_t &out_key)
{
    out_key = 0;
    if (nk > FSV_DG_K) return -1;
    for (int z = 1; z < nk; z++) { const uint32_t kv = D.keys[z]; int z2 = z; for (; z2 > 0 && D.keys[z2 - 1] > kv; z2--) D.keys[z2] = D.keys[z2 - 1]; D.keys[z2] = kv; }
    uint32_t distinct[FSV_DG_D]; int cnt[FSV_DG_D], nd = 0;
    for (int i = 0; i < nk; i++) {
        if (nd && distinct[nd - 1] == D.keys[i]) { cnt[nd - 1]++; continue; }
        if (nd == FSV_DG_D) return -1;
        distinct[nd] = D.keys[i]; cnt[nd] = 1; nd++;
    }
    if (nd == 1) { out_key = distinct[0]; return cnt[0]; }      // one string: the chain itself
    D.n_node = 0; D.n_edge = 0; D.ok = 1;
    const int S = dg_node(D, 'S'), E = dg_node(D, 'E');
    for (int i = 0; i < nd; i++) {
        const int len = (int)(distinct[i] >> 24);
        int last = S;
        for (int j = 0; j < len; j++) { const int nn = dg_node(D, (uint8_t)("ACGT"[(distinct[i] >> (2 * j)) & 3u])); dg_edge(D, last, nn, cnt[i], 0); last = nn; }
        if (last != S) dg_edge(D, last, E, cnt[i], 0);
    }
    int qh = 0, qt = 0;
    D.queue[qt++] = (uint8_t)S;
    while (qh < qt && D.ok) {
        const int cur = D.queue[qh++];
        dg_merge<false>(D, cur);
        dg_merge<true>(D, cur);
        if (!D.alive[cur]) continue;
        for (int i = 0; i < D.out_n[cur]; i++) if (D.e_alive[D.out_e[cur][i]]) D.e_vis[D.out_e[cur][i]] = 1;
        for (int i = 0; i < D.out_n[cur]; i++) {
            const int e = D.out_e[cur][i];
            if (!D.e_alive[e]) continue;
            const int o = D.e_to[e];
            bool all = true;
            for (int j = 0; j < D.in_n[o]; j++) if (D.e_alive[D.in_e[o][j]] && !D.e_vis[D.in_e[o][j]]) { all = false; break; }
            if (all) { if (qt < 4 * FSV_DG_E) D.queue[qt++] = (uint8_t)o; else D.ok = 0; }
        }
    }
    if (!D.ok) return -1;
    int best_s = -1, best_e = -1, ws = 0, we = 0;
    for (int i = 0; i < D.out_n[S]; i++) { const int e = D.out_e[S][i]; if (D.e_alive[e]) { const int o = D.e_to[e], w = o == E ? dg_weight(D, E, true) : dg_weight(D, o, false); if (w > ws) { ws = w; best_s = o; } } }
    for (int i = 0; i < D.in_n[E]; i++) { const int e = D.in_e[E][i]; if (D.e_alive[e]) { const int o = D.e_from[e], w = dg_weight(D, o, false); if (w > we) { we = w; best_e = o; } } }
    uint32_t seq = 0; int L = 0;
    if (ws >= we) {
        int cur = best_s;
        while (cur >= 0 && cur != E && L < FSV_DG_N) {
            int mx = 0, nx = -1;
            if (L < FSV_INS_MAXLEN) seq |= (uint32_t)(D.base[cur] == 'A' ? 0u : D.base[cur] == 'C' ? 1u : D.base[cur] == 'G' ? 2u : 3u) << (2 * L);
            L++;
            for (int i = 0; i < D.out_n[cur]; i++) { const int e = D.out_e[cur][i]; if (D.e_alive[e]) { const int o = D.e_to[e], w = o == E ? dg_weight(D, E, true) : dg_weight(D, o, false); if (w > mx) { mx = w; nx = o; } } }
            cur = nx;
        }
    } else {
        // backward: the bases come out last first
        uint8_t rb[FSV_INS_MAXLEN + 4];
        int cur = best_e;
        while (cur >= 0 && cur != S && L < FSV_DG_N) {
            int mx = 0, nx = -1;
            if (L < FSV_INS_MAXLEN + 4) rb[L] = D.base[cur];
            L++;
            for (int i = 0; i < D.in_n[cur]; i++) { const int e = D.in_e[cur][i]; if (D.e_alive[e]) { const int o = D.e_from[e], w = dg_weight(D, o, false); if (w > mx) { mx = w; nx = o; } } }
            cur = nx;
        }
        const int Lc = min(L, FSV_INS_MAXLEN + 4);
        for (int i = 0; i < Lc && i < FSV_INS_MAXLEN; i++) { const uint8_t b = rb[Lc - 1 - i]; seq |= (uint32_t)(b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : 3u) << (2 * i); }
    }
    if (L > FSV_INS_MAXLEN) L = FSV_INS_MAXLEN;
    out_key = ((uint32_t)L << 24) | seq;
    return ws >= we ? ws : we;
}

// get_seq_from_Graph (Correct.cpp:4010-4129) at the node in front of one backbone column, as oracle/asm.c:vote_consensus: the
// edges to the four bases (the backbone's own first; weight minus the votes that arrived "after an insertion" while the node still
// has insertions to place), the inserted strings' DAG, the deletion edge; the heaviest wins if it has 60 % of the total (51.5 % when
// the PREVIOUS backbone base sits in a homopolymer run); an insertion is written and the node looked at again without it.
// W[b]: votes for base b (the backbone's own + 1), Ifl[b]: of those, votes whose previous cigar run was an insertion, dl: votes
// without a partner for the column, ni: overlaps inserting in front of it, (mi, ikey): the DAG's answer.  out[0] = bases written.
__device__ __forceinline__ bool poa_decide(const int W[4], const int Ifl[4], int dl, int ni, int mi, uint32_t ikey, int own, bool homo, uint8_t *out)
{
    uint8_t nb = 0;
    bool kept = true;
    for (int visit = 0; visit < 2; visit++) {
        int maxc = -1, type = 0, edge = own, total = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int b = k == 0 ? own : (k - 1 < own ? k - 1 : k);     // own first, then the others in base order
            if (W[b] == 0) continue;
            const int cw = ni ? W[b] - Ifl[b] : W[b];
            total += cw;
            if (cw > maxc) { maxc = cw; type = 0; edge = b; }
        }
        if (ni) { total += ni; if (mi > maxc) { maxc = mi; type = 1; } }
        if (dl) { total += dl; if (dl > maxc) { maxc = dl; type = 2; } }
        if (maxc * 5 >= total * 3 || (homo && maxc * 1000 >= total * 515)) {
            if (type == 1) { const int L = (int)(ikey >> 24); for (int b = 0; b < L; b++) out[1 + nb++] = (uint8_t)((ikey >> (2 * b)) & 3u); ni = 0; continue; }
            if (type == 2) { kept = false; break; }
            out[1 + nb++] = (uint8_t)edge;
            break;
        }
        out[1 + nb++] = (uint8_t)own;
        break;
    }
    out[0] = nb;
    return kept;
}

// the most frequent inserted string of a column's event list, the smaller key on a tie
__device__ __forceinline__ void most_frequent_insertion(const uint32_t *s_evkey, const uint16_t *s_evnext, uint32_t head, int &mi, uint32_t &key)
{
    int bc = 0; uint32_t bk = 0;
    for (uint32_t i = head; i != 0xffffu; i = s_evnext[i]) {
        const uint32_t k1 = s_evkey[i];
        int cn = 0;
        for (uint32_t j2 = head; j2 != 0xffffu; j2 = s_evnext[j2]) cn += (s_evkey[j2] == k1);
        if (cn > bc || (cn == bc && k1 < bk)) { bc = cn; bk = k1; }
    }
    mi = bc; key = bk;
}

// the insertion consensus of every column that has insertions, by lane 0 (a handful per window): the column's keyed events are
// gathered from its list, the DAG (or, beyond its bounds, the most frequent string) answers, and the answer replaces the list's head
// event (key, count); bit 16 of the head marks the column as answered
#define FSV_INSLIST 128
#define COV_LO(v) ((int)(int16_t)((v) & 0xffff))
#define COV_HI(v) (((int)(v) - COV_LO(v)) >> 16)
template <int EVC>
__device__ __forceinline__ void answer_insertions(DagLds &D, uint32_t *s_evhead, uint32_t *s_evkey, uint16_t *s_evnext, const uint16_t *s_inslist, int n_list, int n_cols)
{
    const int n = n_list > FSV_INSLIST ? n_cols : n_list;      // the list overflowed: every column
    for (int li = 0; li < n; li++) {
        const int c = n_list > FSV_INSLIST ? li : (int)s_inslist[li];
        const uint32_t head = s_evhead[c] & 0xffffu;
        if (head == 0xffffu || (s_evhead[c] & 0x10000u)) continue;
        int nk = 0;
        for (uint32_t i = head; i != 0xffffu; i = s_evnext[i]) { if (nk < FSV_DG_K) D.keys[nk] = s_evkey[i]; nk++; }
        uint32_t key = 0;
        int mi = dag_insertion(D, nk, key);
        if (mi < 0) most_frequent_insertion(s_evkey, s_evnext, head, mi, key);
        s_evkey[head] = key; s_evnext[head] = (uint16_t)mi;
        s_evhead[c] |= 0x10000u;
    }
}

// if_is_homopolymer_strict (Correct.h:447-530) on 2-bit bases: the run that starts right after the site and the run that starts right
// before it, each looked at over at most three bases; the site joins the forward run if it has that base, else the backward run if
// it has that one; a run of three (the site included or merely beside it) makes a homopolymer site, and so do a forward and a
// backward run of the site's own base that add up to three.  B(p): base at read position p (0 <= p < len).
template <class F>
__device__ __forceinline__ bool homo_strict(F B, int site, int len)
{
    const int beg = max(0, site - 3), end = min(len - 1, site + 3);
    const uint32_t own = B(site);
    uint32_t f_ch = 4u, b_ch = 4u;      // 4: no base seen
    int f_len = 0, b_len = 0;
    for (int i = site + 1; i <= end; i++) {
        const uint32_t v = B(i);
        if (f_ch == 4u) { f_ch = v; f_len = 1; } else if (v != f_ch) break; else f_len++;
    }
    for (int i = site - 1; i >= beg; i--) {
        const uint32_t v = B(i);
        if (b_ch == 4u) { b_ch = v; b_len = 1; } else if (v != b_ch) break; else b_len++;
    }
    if (f_ch == own) f_len++;
    else if (b_ch == own) b_len++;
    return f_len >= 3 || b_len >= 3 || (own == f_ch && b_ch == f_ch && f_len + b_len >= 3);
}

// inclusive prefix sum over the wavefront's lanes
__device__ __forceinline__ int wave_incl_sum(int v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(v, off, 64); if (lane >= off) v += o; }
    return v;
}
// ---- one overlap's votes on the columns of a window: the walk over its 2-bit path --------------------------------------------------
// A match op votes for the backbone's own base, so a path only contributes its deviations.  The walk jumps from deviation to
// deviation (a bit per non-zero path word, count-trailing-zeros inside a word) and does not touch memory: what needs a base of y
// -- a mismatch's vote, the string of an insertion -- is put aside, four at a time, and the y words of all four are requested
// together (a load per deviating op inside the walk was a dependent memory round trip per deviation for the whole wave).
// row: the lane's 26 op words in LDS (fields past plen are 0), nz: bit w set = word w is not all matches.
// pend: an insertion in front of column xs is already pending (the junction vote).  Returns the number of y-only ops (n2).
#define FSV_PEND_N 4
struct ConsLds {
    uint32_t (*cnt)[3]; int32_t *cov; uint32_t *fl; uint32_t *evn; uint32_t *evkey; uint16_t *evnext; uint32_t *evhead;
};
template <int EVC>
__device__ __forceinline__ void cons_flush(const ConsLds &S, const uint32_t *__restrict__ store, uint32_t y_word, int y_len, int y_rev, int ry_start,
                                           const uint32_t (&ent)[FSV_PEND_N], int n)
{
    // entry: column | (y position - ry_start) << 9 | L << 19 (0: a mismatch) | flagged << 23
    uint32_t bits[FSV_PEND_N];
#pragma unroll
    for (int i = 0; i < FSV_PEND_N; i++) if (i < n) bits[i] = fetch16(store, y_word, y_len, y_rev, ry_start + (int)((ent[i] >> 9) & 1023u)).bits;
#pragma unroll
    for (int i = 0; i < FSV_PEND_N; i++) {
        if (i >= n) continue;
        const uint32_t e = ent[i], xp = e & 511u, L = (e >> 19) & 15u;
        if (L == 0u) {
            const uint32_t yb = bits[i] & 3u;
            atomicAdd(&S.cnt[xp][yb >> 1], 1u << ((yb & 1u) << 4));
            if (e >> 23) atomicAdd(&S.fl[xp], 1u << (yb << 3));
        } else {
            const uint32_t key = (L << 24) | (bits[i] & ((1u << (2u * L)) - 1u));
            const uint32_t ev = atomicAdd(S.evn, 1u);
            if (ev < (uint32_t)EVC) { S.evkey[ev] = key; S.evnext[ev] = (uint16_t)atomicExch(&S.evhead[xp], ev); }
        }
    }
}
template <int EVC>
__device__ __forceinline__ int cons_walk(const ConsLds &S, const uint32_t *row, uint32_t nz, int plen, int xs, int xlim, bool pend,
                                         const uint32_t *__restrict__ store, uint32_t y_word, int y_len, int y_rev, int ry_start)
{
    int n2 = 0, n3 = 0, fstart = -1, fmm_pos = -1, np = 0, p = 0;
    uint32_t ent[FSV_PEND_N] = {0u, 0u, 0u, 0u};
    while (true) {
        if (np == FSV_PEND_N) { cons_flush<EVC>(S, store, y_word, y_len, y_rev, ry_start, ent, np); np = 0; }
        if (!pend) {       // on to the next op that is not a match
            int w = p >> 4;
            if (w >= 26) break;
            uint32_t rest = row[w] >> ((p & 15) << 1);
            if (rest == 0u) {
                const uint32_t m = nz >> (w + 1);
                if (m == 0u) break;
                w += 1 + __builtin_ctz(m);
                p = w << 4;
                rest = row[w];
            }
            p += __builtin_ctz(rest) >> 1;
        }
        if (p >= plen) break;
        const uint32_t op = (row[p >> 4] >> ((p & 15) << 1)) & 3u;
        const int xp = xs + p - n2;
        if (fstart >= 0 && op != 0u) { atomicAdd(&S.cov[fstart], 65536); atomicAdd(&S.cov[xp], -65536); fstart = -1; }   // the match run after an insertion ends here
        if (op == 2u) {     // run of y-only ops in front of column xp
            int L = 1;
            while (p + L < plen && ((row[(p + L) >> 4] >> (((p + L) & 15) << 1)) & 3u) == 2u) L++;
            if (xp < xlim) {
                pend = true;
                if (L <= FSV_INS_MAXLEN) {
                    const uint32_t e = (uint32_t)xp | ((uint32_t)(p - n3) << 9) | ((uint32_t)L << 19);
                    if (np == 0) ent[0] = e; else if (np == 1) ent[1] = e; else if (np == 2) ent[2] = e; else ent[3] = e;
                    np++;
                }
            }
            n2 += L; p += L;
            continue;
        }
        const bool after_ins = pend && p > 0;      // (a junction vote is no cigar run)
        if (pend) { atomicAdd(&S.cnt[xp][2], 1u << 16); pend = false; }
        if (op == 3u) { atomicAdd(&S.cnt[xp][2], 1u); n3++; }
        else if (op == 1u) {
            // add_mismatchEdge_weight (POA.h:492) looks at the previous cigar RUN: every base of the run that follows an insertion
            // counts as "after an insertion", not only the first
            const bool flagged = after_ins || fmm_pos == p;
            const uint32_t e = (uint32_t)xp | ((uint32_t)(p - n3) << 9) | (flagged ? 1u << 23 : 0u);
            if (np == 0) ent[0] = e; else if (np == 1) ent[1] = e; else if (np == 2) ent[2] = e; else ent[3] = e;
            np++;
            if (flagged) fmm_pos = p + 1;
        } else if (after_ins) fstart = xp;
        p++;
    }
    if (fstart >= 0) { atomicAdd(&S.cov[fstart], 65536); atomicAdd(&S.cov[xs + plen - n2], -65536); }
    if (np) cons_flush<EVC>(S, store, y_word, y_len, y_rev, ry_start, ent, np);
    return n2;
}

struct SiteLists {
    uint32_t *site_cnt;        // per grid window: FSV_SITE_MARK after k_consensus<., 1>, the number of kept sites after k_snp_sites
    uint32_t *win_list;        // marked windows, [0] of win_n
    uint32_t *redo_list;       // windows of the reads that lost an overlap to the partition
    uint32_t *win_n;           // {marked windows, redo windows}
    // per grid window, from k_bcig_accept (nullptr: no junction cigars anywhere): [0] a used junction cigar, or a window cigar it stands in
    // for, has a mismatch in the window; [1] / [2] first / last column at which a used junction cigar shows something else than the
    // window cigar ([1] > [2]: none)
    const int32_t *bc_win;
};
template <int EVC, int MODE>
__device__ __forceinline__ void consensus_window(const ConsArgs &A, const uint32_t gw, const SiteLists &L)
{
    // Per-column votes of one 375-bp grid window.  A match op votes for the backbone's own base, so a lane (= one
    // overlap) only contributes (a) its coverage interval, through a difference array, and (b) its deviations --
    // mismatches, deleted columns, insertions -- which it finds by skipping the all-match words of its 2-bit path.
    // Deviations are sparse (HiFi: ~1.5 per window), so their LDS atomics do not collide the way per-step votes would.
    __shared__ uint32_t s_cnt[FSV_WINDOW + 1][3];  // per column, 16 bits each: votes for A C | G T that differ from the backbone | deleted, arrived-after-insertion
    __shared__ int32_t s_cov[FSV_WINDOW + 2];      // coverage difference array -> arrived
    __shared__ uint32_t s_path[64][27];            // per lane: the 26 op words of its window path (odd stride); reused as s_out
    __shared__ uint16_t s_evnext[EVC];             // insertion events of a column form a list: s_evhead[column] -> event -> s_evnext[event] ...
    __shared__ uint32_t s_evkey[EVC];
    __shared__ uint32_t s_evhead[FSV_WINDOW + 1];
    __shared__ uint32_t s_evn, s_cover, s_anydev;   // s_anydev: some overlap deviates from the backbone somewhere in this window
    // s_cov's upper halves: difference array of the match runs that follow an insertion (the votes hifiasm counts as "after an insertion")
    __shared__ uint32_t s_fl[FSV_WINDOW + 1];      // the same for mismatch runs: per column, 8 bits per base
    __shared__ uint16_t s_inslist[FSV_INSLIST];    // columns whose inserted strings disagree
    __shared__ uint32_t s_nins;
    __shared__ uint16_t s_devlist[FSV_WINDOW + 1]; // columns some vote deviates at
    __shared__ uint32_t s_ndev;
    static_assert(sizeof(DagLds) <= sizeof(uint32_t) * 64 * 27, "the DAG scratch lives in the path buffer between the tally and the decisions");
    DagLds &s_dag = *reinterpret_cast<DagLds *>(&s_path[0][0]);
    __shared__ uint32_t s_xraw[28];                // raw store words covering x[gs-16 .. gs+glen+16)
    const int lane = threadIdx.x;
    const uint4 gt = A.gwin_tab[gw];
    const uint32_t r = gt.x, pbase = gt.y, n_ovl = gt.z;
    const int g = (int)gt.w;
    const int xlen = A.read_len[r];
    const int gs = g * FSV_WINDOW, glen = min(FSV_WINDOW, xlen - gs);
    const uint32_t xw = A.word_off[r];
    const int xw0 = (gs >> 4) - 1; // first staged word (may be -1 at the read start: reads as 0, never used)
    if (lane < 28) { const int wi = xw0 + lane; s_xraw[lane] = (wi >= 0 && wi <= ((xlen + 15) >> 4)) ? A.store[xw + wi] : 0u; }
    if (A.read_dirty && !A.read_dirty[r]) {
        // every accepted overlap of this read matches it base for base (from the second round on: most reads): all votes are for
        // the backbone, whatever the coverage
        __syncthreads();
        uint8_t *dst0 = A.cwin + (size_t)gw * FSV_CW_STRIDE;
        for (int c = lane; c < glen; c += 64) dst0[c] = (uint8_t)((s_xraw[((gs + c) >> 4) - xw0] >> (((gs + c) & 15) << 1)) & 3u);
        if (lane == 0) { A.cwin_len[gw] = (uint16_t)glen; if (MODE == 1) L.site_cnt[gw] = 0u; }
        return;
    }
    for (int i = lane; i < (FSV_WINDOW + 1) * 3; i += 64) (&s_cnt[0][0])[i] = 0;
    for (int i = lane; i < FSV_WINDOW + 2; i += 64) s_cov[i] = 0;
    for (int i = lane; i < FSV_WINDOW + 1; i += 64) s_evhead[i] = 0xffffu;
    for (int i = lane; i < FSV_WINDOW + 1; i += 64) s_fl[i] = 0;
    if (lane == 0) { s_evn = 0; s_cover = 0; s_anydev = 0; s_nins = 0; s_ndev = 0; }
    __syncthreads();
    const ConsLds CL = {s_cnt, s_cov, s_fl, &s_evn, s_evkey, s_evnext, s_evhead};
#define XB(p) ((s_xraw[((p) >> 4) - xw0] >> (((p) & 15) << 1)) & 3u)
#define CNT_ADD(c, b) atomicAdd(&s_cnt[(c)][(b) >> 1], 1u << (((b) & 1u) << 4))
#define CNT_GET(c, b) ((s_cnt[(c)][(b) >> 1] >> (((b) & 1u) << 4)) & 0xffffu)
    for (uint32_t oi = lane; oi < n_ovl; oi += 64) {
        const uint4 oc = A.ovl_c[pbase + oi];
        const int o_x_s = (int)oc.x, o_n_win = (int)(oc.z & 0x7fffffffu);
        const int j = g - o_x_s / FSV_WINDOW;
        if (!(oc.z >> 31) || j < 0 || j >= o_n_win) continue;
        const uint32_t ti = oc.y + (uint32_t)j;
        const fsv_wpath *P = A.paths + ti;
        const uint4 h0 = *reinterpret_cast<const uint4 *>(P);                      // ry_start, ry_end, path_len|err, state|rev|pad
        const uint2 h1 = *reinterpret_cast<const uint2 *>((const uint8_t *)P + 16); // y_word, y_len
        // In a read somebody deviates from (the only reads that get here) most paths carry deviations, so the 104 op bytes are
        // requested together with the header: waiting for the header first to learn whether the path is clean cost a second
        // memory round trip per overlap on the window's critical path
        uint2 pv[13];
        {
            const uint2 *src = reinterpret_cast<const uint2 *>(P->ops);
#pragma unroll
            for (int i = 0; i < 13; i++) pv[i] = src[i];
        }
        atomicAdd(&s_cover, 1u);       // get_available_interval (Correct.cpp:113): every accepted overlap that overlaps the window counts, matched there or not
        if ((h0.w & 0xffu) != 1u) continue;
        // a path at distance 0 is all matches: it only adds its coverage interval
        const bool clean_path = (int16_t)(h0.z >> 16) == 0;
        uint32_t nz = 0;
        if (!clean_path) {
#pragma unroll
            for (int i = 0; i < 13; i++) {
                s_path[lane][2 * i] = pv[i].x; s_path[lane][2 * i + 1] = pv[i].y;
                nz |= (pv[i].x ? 1u << (2 * i) : 0u) | (pv[i].y ? 2u << (2 * i) : 0u);
            }
        }
        const int ry_start = (int)h0.x, plen = (int)(int16_t)(h0.z & 0xffffu);
        const uint32_t y_word = h1.x; const int y_len = (int)h1.y, y_rev = (int)((h0.w >> 8) & 0xffu);
        const int xs = max(gs, o_x_s) - gs;
        bool pend = false;
        if (j > 0 && A.junction_vote) {
            const uint4 hp = *reinterpret_cast<const uint4 *>(A.paths + ti - 1);
            if ((hp.w & 0xffu) == 1u) {
                const int gap = ry_start - (int)hp.y - 1;
                if (gap > 0 && xs == 0) {
                    pend = true;
                    if (gap <= FSV_INS_MAXLEN) {
                        uint32_t key = (uint32_t)gap << 24;
                        for (int b = 0; b < gap; b++) key |= fsv_base_at(A.store, y_word, y_len, y_rev, ry_start - gap + b) << (2 * b);
                        uint32_t e = atomicAdd(&s_evn, 1u);
                        if (e < (uint32_t)EVC) { s_evkey[e] = key; s_evnext[e] = (uint16_t)atomicExch(&s_evhead[0], e); }
                    }
                }
            }
        }
        if (pend || nz) s_anydev = 1u;
        int n2 = 0;
        if (clean_path) { if (pend) CNT_ADD(xs, 5u); }      // the first op is a match at column xs
        else n2 = cons_walk<EVC>(CL, s_path[lane], nz, plen, xs, glen, pend, A.store, y_word, y_len, y_rev, ry_start);
        // coverage interval: every x base of the task is consumed exactly once
        atomicAdd(&s_cov[xs], 1);
        atomicAdd(&s_cov[xs + plen - n2], -1);
    }
    __syncthreads();
    // arrived[c] = prefix sum of the difference array; each lane owns the contiguous columns [c0, c1)
    const int per = (glen + 63) / 64, c0 = min(glen, lane * per), c1 = min(glen, c0 + per);
    // (low half: coverage, high half: the flagged match runs; both small signed numbers, so the halves separate exactly)
    int run = 0, frun = 0;
    for (int c = c0; c < c1; c++) {
        const int v = s_cov[c];
        run += COV_LO(v); frun += COV_HI(v);
        if (CNT_GET(c, 5u)) {       // inserted strings that disagree go through the DAG (lane 0, below); one string answers itself
            const uint32_t head = s_evhead[c];
            bool same = true;
            if (head != 0xffffu) { const uint32_t k0 = s_evkey[head]; for (uint32_t i = s_evnext[head]; i != 0xffffu; i = s_evnext[i]) if (s_evkey[i] != k0) { same = false; break; } }
            if (!same && A.ins_dag) { const uint32_t k = atomicAdd(&s_nins, 1u); if (k < FSV_INSLIST) s_inslist[k] = (uint16_t)c; }
        }
    }
    const int before = wave_incl_sum(run, lane) - run, fbefore = wave_incl_sum(frun, lane) - frun;
    __syncthreads();
    if (lane == 0 && s_nins) answer_insertions<EVC>(s_dag, s_evhead, s_evkey, s_evnext, s_inslist, (int)s_nins, glen);
    __syncthreads();
    uint8_t (*s_out)[14] = reinterpret_cast<uint8_t (*)[14]>(&s_path[0][0]); // 375 x 14 B = 5.2 KB <= 64 x 27 x 4 B; paths are done
    uint8_t *dst = A.cwin + (size_t)gw * FSV_CW_STRIDE;
    // fewer than three overlaps: the reference leaves the window alone; no deviation anywhere: every vote is for the backbone
    const bool verbatim = s_cover < 3u || s_anydev == 0u;
    if (A.cov3 && lane == 0) A.cov3[gw] = s_cover >= 3u ? 1 : 0;
    if (s_evn > (uint32_t)EVC && lane == 0) atomicOr(&A.warn[r], (uint32_t)FSV_W_INS_EVENTS);
    // Columns nobody deviates at (no vote in s_cnt: nine in ten even in the first round) keep the backbone's base: poa_decide then
    // sees one edge with all the weight.  The others are listed and decided a lane each -- walked in place, lane = six consecutive
    // columns, nearly every trip had some lane with a deviating column and the whole wave went through the decision six times.
    int arrived = before, farrived = fbefore;
    bool differs = false, site = false;
    const int bc_lo = (MODE == 1 && L.bc_win) ? L.bc_win[3 * (size_t)gw + 1] : 1, bc_hi = (MODE == 1 && L.bc_win) ? L.bc_win[3 * (size_t)gw + 2] : 0;
    for (int c = c0; c < c1; c++) {
        arrived += COV_LO(s_cov[c]);
        farrived += COV_HI(s_cov[c]);
        s_out[c][0] = 1; s_out[c][1] = (uint8_t)XB(gs + c);
        if (!verbatim && (s_cnt[c][0] | s_cnt[c][1] | s_cnt[c][2]) != 0u) {
            s_cov[c] = (int32_t)(((uint32_t)arrived & 0xffffu) | ((uint32_t)farrived << 16));     // the lane owns its columns: the difference array is done with
            s_devlist[atomicAdd(&s_ndev, 1u)] = (uint16_t)c;
        }
    }
    __syncthreads();
    for (uint32_t e = lane; e < s_ndev; e += 64) {
        const int c = (int)s_devlist[e];
        arrived = (int)(int16_t)((uint32_t)s_cov[c] & 0xffffu); farrived = (int)(int16_t)((uint32_t)s_cov[c] >> 16);
        const uint32_t own = XB(gs + c);
        uint8_t nb = 0;
        if (MODE == 1) {
            // split_sub_list (Correct.cpp:5804) on the same tallies, as in k_snp_sites
            int oa[4], occ1 = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) { oa[b] = (int)CNT_GET(c, (uint32_t)b); occ1 += oa[b]; }
            // (a column with two mismatch votes where some overlap's junction cigar shows something else than its window cigar: the
            // tallies of the partition differ from these -- k_snp_sites decides)
            if (occ1 > 1 && c >= bc_lo && c <= bc_hi) site = true;
            if (occ1 > 1) {
                const int occ2 = (int)CNT_GET(c, 4u), occ0 = arrived - occ1 - occ2;
                int mx = occ2, mi = -1;
#pragma unroll
                for (int b = 0; b < 4; b++) if (oa[b] > mx) { mx = oa[b]; mi = b; }
                bool ok = occ0 != 0 && mi >= 0 && mx > 1;
#pragma unroll
                for (int b = 0; b < 4; b++) if (oa[b] == mx && b != mi) ok = false;
                if (ok && (double)(occ0 + 1 + mx) / (double)(arrived + 1) >= 0.95 && (double)mx / (double)(arrived + 1 - (occ0 + 1)) >= 0.70) site = true;
            }
        }
        {
            // the node in front of column c (poa_decide); the homopolymer relief looks at the PREVIOUS backbone base
            const bool homo = c > 0 && homo_strict([&](int pp) { return XB(pp); }, gs + c - 1, xlen);
            int W[4], Ifl[4], dev = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) { W[b] = (int)CNT_GET(c, (uint32_t)b); dev += W[b]; Ifl[b] = (int)((s_fl[c] >> (b << 3)) & 0xffu); }
            const int dl = (int)CNT_GET(c, 4u), ni = (int)CNT_GET(c, 5u);
#pragma unroll
            for (int b = 0; b < 4; b++) if ((int)own == b) { W[b] += arrived - dev - dl + 1; Ifl[b] = farrived; }
            int mi = 0; uint32_t ikey = 0;
            if (ni) {
                const uint32_t hv = s_evhead[c], head = hv & 0xffffu;
                if (head != 0xffffu) {
                    ikey = s_evkey[head];
                    if (hv & 0x10000u) mi = (int)s_evnext[head];                                   // the DAG's answer
                    else if (A.ins_dag) { for (uint32_t i = head; i != 0xffffu; i = s_evnext[i]) mi++; }   // one string, mi times
                    else most_frequent_insertion(s_evkey, s_evnext, head, mi, ikey);
                }
            }
            poa_decide(W, Ifl, dl, ni, mi, ikey, (int)own, homo, &s_out[c][0]);
            nb = s_out[c][0];
        }
        if (nb != 1 || s_out[c][1] != (uint8_t)own) differs = true;
    }
    if (differs && A.changed) A.changed[r] = 1u;
    if (MODE == 1) {
        // a window a used junction cigar reaches into is looked at by k_snp_sites whatever the window cigars say: what its overlaps
        // show beside the junction is read off that cigar there (markSNP_advance, Correct.cpp:5054)
        const bool any_site = __ballot(site) != 0ull || (L.bc_win && L.bc_win[3 * (size_t)gw]);
        if (lane == 0) {
            L.site_cnt[gw] = any_site ? FSV_SITE_MARK : 0u;
            if (any_site) L.win_list[atomicAdd(&L.win_n[0], 1u)] = gw;
        }
    }
    __syncthreads();
    uint32_t mine = 0;
    for (int c = c0; c < c1; c++) mine += s_out[c][0];
    const int incl = wave_incl_sum((int)mine, lane);
    uint32_t off = (uint32_t)incl - mine;
    const uint32_t tot = (uint32_t)__shfl(incl, 63, 64);
    if (tot > FSV_CW_STRIDE) { // cannot happen with <= 12-base insertions winning at a few columns; keep the read as it is
        for (int c = lane; c < glen; c += 64) dst[c] = (uint8_t)XB(gs + c);
        if (lane == 0) { A.cwin_len[gw] = (uint16_t)glen; atomicOr(&A.warn[r], (uint32_t)FSV_W_WINDOW_KEPT); }
        return;
    }
    for (int c = c0; c < c1; c++) for (int b = 0; b < s_out[c][0]; b++) dst[off++] = s_out[c][1 + b];
    if (lane == 0) A.cwin_len[gw] = (uint16_t)tot;
#undef XB
#undef CNT_ADD
#undef CNT_GET
}

template <int EVC, int MODE>
__global__ __launch_bounds__(64) void k_consensus(ConsArgs A, uint32_t n_gwin, SiteLists L)
{
    uint32_t gw;
    if (xcd_block(n_gwin, gw)) consensus_window<EVC, MODE>(A, gw, L);
}

// the redo: a fixed grid walks the list k_hap_partition left
template <int EVC>
__global__ __launch_bounds__(64) void k_consensus_redo(ConsArgs A, SiteLists L)
{
    const uint32_t n = L.win_n[1];
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        consensus_window<EVC, 0>(A, L.redo_list[i], L);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ k_bcig_tasks / k_bcig_accept
// calculate_boundary_cigars (Correct.cpp:2310-2530), for the haplotype partition: the junction between two matched windows of an
// accepted overlap whose alignments do not simply meet (bases of y skipped or used twice, or an error within 10 columns of the
// junction on either side) is aligned once more -- up to 100 columns on each side, doubled threshold, no fix_boundary -- and the
// partition reads the ~50 columns on each side of the junction off that cigar unless it has clearly more errors there than the two
// window cigars (markSNP_advance :5054, addSNPtohaplotype_advance :5351).  k_bcig_tasks writes the junction tasks (one thread per
// window task; K5 and the K6 kernels then run on them as on any task list), k_bcig_accept decides which of the new cigars are used.
// oracle/asm.c:boundary_cigars / window_evidence are the same, statement for statement.
#define FSV_BC_SIDE 100
#define FSV_BC_USELESS 50
#define FSV_BC_SCAN 10
struct BcigArgs {
    const fsv_wtask *tasks; const fsv_wpath *paths; const uint32_t *n_tasks;     // the round's window tasks and their paths
    const uint32_t *pair_read, *read_dirty, *gwin_off; const uint8_t *thr_tab; int k_cap;
    fsv_wtask *tasks2; int32_t *bc_idx; uint32_t *n_tasks2;                      // junction tasks; bc_idx[window task] = the task of the junction behind it or -1
    const fsv_wres *res2; const fsv_wpath *paths2; uint4 *bc_rec; int32_t *bc_win;      // bc_win: see SiteLists
    uint32_t *n_same, *n_used;      // statistics: accepted cigars that show what the window cigars show / that are used
};
// op i of a path record's 2-bit stream, through one cached word
struct OpReader {
    const uint32_t *w; uint32_t cur = 0; int idx = -1;
    __device__ __forceinline__ OpReader(const fsv_wpath *P) : w(reinterpret_cast<const uint32_t *>(P->ops)) {}
    __device__ __forceinline__ uint32_t get(int i) { const int wi = i >> 4; if (wi != idx) { cur = w[wi]; idx = wi; } return (cur >> ((i & 15) << 1)) & 3u; }
};
// scan_cigar (Correct.cpp:1070-1200): errors met while the first (dir 0) / last (dir 1) scan_x columns of x go by; y-only ops count
// whenever they are met.  err0: the path's distance (a distance-0 record carries no ops)
__device__ __forceinline__ int scan_ops(const fsv_wpath *P, int plen, int err0, int scan_x, int dir)
{
    if (err0 == 0) return 0;
    OpReader R(P);
    int x_i = 0, err = 0;
    for (int p = 0; p < plen; p++) {
        const uint32_t op = R.get(dir ? plen - 1 - p : p);
        if (op == 2u) { err++; continue; }
        if (op != 0u) err++;
        if (++x_i >= scan_x) return err;
    }
    return err;
}
// scan_cigar_interval (Correct.cpp:1204-1290): errors over the columns [xb, xe] of x
__device__ __forceinline__ int scan_ops_interval(const fsv_wpath *P, int plen, int err0, int xb, int xe)
{
    if (err0 == 0) return 0;
    OpReader R(P);
    int x_i = 0, err = 0;
    for (int p = 0; p < plen; p++) {
        const uint32_t op = R.get(p);
        if (op == 2u) { err++; continue; }
        if (x_i == xb) err = 0;
        x_i++;
        if (op != 0u) err++;
        if (x_i == xe + 1) return err;
    }
    return err;
}

__global__ void k_bcwin_init(int32_t *__restrict__ bc_win, uint32_t n_gwin)
{
    const uint32_t gw = blockIdx.x * blockDim.x + threadIdx.x;
    if (gw < n_gwin) { bc_win[3 * (size_t)gw] = 0; bc_win[3 * (size_t)gw + 1] = 0x7fffffff; bc_win[3 * (size_t)gw + 2] = -1; }
}

__global__ __launch_bounds__(256) void k_bcig_tasks(BcigArgs A)
{
    const uint32_t n_tasks = min(*A.n_tasks, gridDim.x * blockDim.x);
    uint32_t blk;
    if (!xcd_block((n_tasks + 255u) >> 8, blk)) return;
    const uint32_t ti = blk * blockDim.x + threadIdx.x;
    if (ti >= n_tasks) return;
    A.bc_idx[ti] = -1;
    if (ti + 1 >= n_tasks) return;
    // (the two headers say everything about the junction but where it lies: loaded together with the task, one round trip)
    const fsv_wtask t0 = A.tasks[ti];
    // a clean read (most reads from the second round on): every path at distance 0, every junction met -- its path records (a 128-byte
    // line each for 16 bytes of header) are not even looked at
    if (!A.read_dirty[A.pair_read[t0.ovl]]) return;
    const uint4 h0 = *reinterpret_cast<const uint4 *>(A.paths + ti), h1 = *reinterpret_cast<const uint4 *>(A.paths + ti + 1);
    if ((h0.w & 0xffu) != 1u || (h1.w & 0xffu) != 1u) return;     // (a path only exists for a matched window of an accepted overlap)
    const int y_distance = (int)h1.x - (int)h0.y - 1;
    // nothing to re-align where the two alignments meet and neither shows an error within ten columns of the junction
    if (y_distance == 0 && !((h0.w >> 16) & 2u) && !((h1.w >> 16) & 1u)) return;
    const fsv_wtask t1 = A.tasks[ti + 1];
    if (t1.ovl != t0.ovl) return;                           // the overlap's last window
    int y_start = (int)h0.y, x_start = t0.x_start + (int)t0.x_len - 1;
    const int leftLen = min(min(x_start - t0.x_start, y_start), FSV_BC_SIDE);
    const int rightLen = min(min(t1.x_start + (int)t1.x_len - x_start, t0.y_len - y_start), FSV_BC_SIDE);
    const int xLen = leftLen + rightLen;
    if (xLen <= 0) return;
    x_start -= leftLen; y_start -= leftLen;
    const int thr = double_thr(A.thr_tab[xLen], xLen, A.k_cap);
    if (thr > FSV_K_MAX) return;
    const uint32_t slot = atomicAdd(A.n_tasks2, 1u);
    fsv_wtask w;
    w.x_word = t0.x_word; w.y_word = t0.y_word; w.x_start = x_start; w.y_start = y_start; w.y_len = t0.y_len;
    w.x_len = (uint16_t)xLen; w.k = (uint8_t)thr; w.y_rev = t0.y_rev; w.ovl = t0.ovl; w.win = ti;
    A.tasks2[slot] = w;
    A.bc_idx[ti] = (int32_t)slot;
}

// one thread per junction task: is its cigar used?  Not when it has clearly more errors in its inner columns than the two window
// cigars have there; and not when it shows, column for column, what the window cigars show (then nothing changes and the windows
// beside it need no second look by k_snp_sites -- most of them: the re-alignment usually finds the two window alignments again)
__global__ __launch_bounds__(256) void k_bcig_accept(BcigArgs A)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= *A.n_tasks2) return;
    A.bc_rec[slot] = make_uint4(0, 0, 0, 0);
    const fsv_wtask w = A.tasks2[slot];
    const fsv_wres r = A.res2[slot];
    const uint32_t ti = w.win;
    const fsv_wpath *PB = A.paths2 + slot, *P0 = A.paths + ti, *P1 = A.paths + ti + 1;
    const uint4 hb = *reinterpret_cast<const uint4 *>(PB);
    if (r.err < 0 || (hb.w & 0xffu) != 1u) return;
    const int xLen = w.x_len, thr = w.k;
    if (xLen + 2 * thr - r.extra_begin - r.extra_end < xLen) return;      // o_len < xLen
    const fsv_wtask t0 = A.tasks[ti], t1 = A.tasks[ti + 1];
    const uint4 h0 = *reinterpret_cast<const uint4 *>(P0), h1 = *reinterpret_cast<const uint4 *>(P1);
    const int x_end0 = t0.x_start + (int)t0.x_len - 1, leftLen = x_end0 - w.x_start, rightLen = xLen - leftLen;
    int y_distance = (int)h1.x - (int)h0.y - 1;
    if (y_distance < 0) y_distance = -y_distance;
    int L = FSV_BC_USELESS, R = FSV_BC_USELESS;
    const uint32_t first_ti = ti - t0.win;                  // (window tasks carry their index inside the overlap)
    if (t0.win == 0 && w.x_start == t0.x_start) L = 0;
    {
        // the overlap's last window: the next task belongs to another overlap (or there is none)
        const bool last_junction = ti + 2 >= *A.n_tasks || A.tasks[ti + 2].ovl != t0.ovl;
        if (last_junction && w.x_start + xLen - 1 == t1.x_start + (int)t1.x_len - 1) R = 0;
    }
    (void)first_ti;
    if (leftLen <= L || rightLen <= R) return;
    const int plb = (int)(int16_t)(hb.z & 0xffffu), eb = (int)(int16_t)(hb.z >> 16);
    const int pl0 = (int)(int16_t)(h0.z & 0xffffu), e0 = (int)(int16_t)(h0.z >> 16), pl1 = (int)(int16_t)(h1.z & 0xffffu), e1 = (int)(int16_t)(h1.z >> 16);
    const int m_err = scan_ops_interval(PB, plb, eb, L, xLen - R - 1);
    const int b_err = scan_ops(P0, pl0, e0, leftLen - L, 1), f_err = scan_ops(P1, pl1, e1, rightLen - R, 0);
    if (f_err + b_err + y_distance + 1 < m_err) return;
    // Does it show anything the window cigars do not?  Column by column over the columns it would be used for: the same op, and
    // for a column with a partner the same base of y (its position).
    bool differs = false, mm0 = false, mm1 = false;
    int lo0 = 0x7fffffff, hi0d = -1, lo1 = 0x7fffffff, hi1d = -1;      // first / last differing junction column on either side
    {
        // columns of the junction cigar that stand for window 0: [L, leftLen] (its last column included); for window 1:
        // [leftLen + 1, leftLen + 1 + (rightLen - 1 - R) - 1] (markSNP_advance's intervals, restated in snp_sites_window)
        OpReader RB(PB), R0(P0), R1(P1);
        int pb = 0, xb = 0, yb = (int)hb.x;
        // window 0 from its column x0c on: skip its ops in front of that column
        const int x0c = (w.x_start + L) - t0.x_start;
        int p0 = 0, x0 = 0, y0 = (int)h0.x;
        if (e0 != 0) { while (p0 < pl0 && x0 < x0c) { const uint32_t op = R0.get(p0++); if (op == 2u) y0++; else { x0++; if (op != 3u) y0++; } } }
        else { x0 = x0c; y0 += x0c; p0 = x0c; }
        if (eb != 0) { while (pb < plb && xb < L) { const uint32_t op = RB.get(pb++); if (op == 2u) yb++; else { xb++; if (op != 3u) yb++; } } }
        else { xb = L; yb += L; pb = L; }
        auto next = [](OpReader &Rd, int &p, int pl, int e, int &y, uint32_t &op_out, int &y_out) {
            // the next op that consumes a column of x: its code and the position of its partner in y
            if (e == 0) { op_out = 0u; y_out = y; y++; p++; return; }
            while (p < pl) { const uint32_t op = Rd.get(p++); if (op == 2u) { y++; continue; } op_out = op; y_out = y; if (op != 3u) y++; return; }
            op_out = 0u; y_out = y;
        };
        const int hi0 = leftLen;                                   // last junction column read for window 0
        for (; xb <= hi0; xb++) {
            uint32_t ob, ow; int ybp, ywp;
            next(RB, pb, plb, eb, yb, ob, ybp);
            next(R0, p0, pl0, e0, y0, ow, ywp);
            if (ob != ow || (ob != 3u && ybp != ywp)) { if (lo0 > xb) lo0 = xb; hi0d = xb; }
            if (ob == 1u || ow == 1u) mm0 = true;
        }
        // window 1: junction columns leftLen + 1 .. leftLen + (rightLen - 1 - R)
        int p1 = 0, y1 = (int)h1.x;
        const int hi1 = leftLen + (rightLen - 1 - R);
        for (; xb <= hi1; xb++) {
            uint32_t ob, ow; int ybp, ywp;
            next(RB, pb, plb, eb, yb, ob, ybp);
            next(R1, p1, pl1, e1, y1, ow, ywp);
            if (ob != ow || (ob != 3u && ybp != ywp)) { if (lo1 > xb) lo1 = xb; hi1d = xb; }
            if (ob == 1u || ow == 1u) mm1 = true;
        }
        differs = lo0 <= hi0d || lo1 <= hi1d;
    }
    if (!differs) { atomicAdd(A.n_same, 1u); return; }
    atomicAdd(A.n_used, 1u);
    A.bc_rec[slot] = make_uint4(1u, (uint32_t)w.x_start, (uint32_t)xLen | ((uint32_t)L << 16) | ((uint32_t)R << 24), 0u);
    // what k_consensus needs to know about the two windows: the columns where the partition's tallies can differ from its own
    const uint32_t rd = A.pair_read[t0.ovl], g0 = A.gwin_off[rd] + (uint32_t)(t0.x_start / FSV_WINDOW);
    const int gs0 = (t0.x_start / FSV_WINDOW) * FSV_WINDOW;
    if (lo0 <= hi0d) {
        if (mm0) A.bc_win[3 * (size_t)g0] = 1;
        atomicMin(&A.bc_win[3 * (size_t)g0 + 1], w.x_start + lo0 - gs0); atomicMax(&A.bc_win[3 * (size_t)g0 + 2], w.x_start + hi0d - gs0);
    }
    if (lo1 <= hi1d) {
        if (mm1) A.bc_win[3 * (size_t)(g0 + 1)] = 1;
        atomicMin(&A.bc_win[3 * (size_t)(g0 + 1) + 1], w.x_start + lo1 - gs0 - FSV_WINDOW); atomicMax(&A.bc_win[3 * (size_t)(g0 + 1) + 2], w.x_start + hi1d - gs0 - FSV_WINDOW);
    }
}

// ------------------------------------------------------------------------------------------------ k_snp_sites / k_hap_partition
// partition_overlaps_advance (Correct.cpp:7127-7206), for every read of every set as hifiasm runs it (it has no notion of a phased
// input: in a phased set the "heterozygous" columns are coincident read errors, and the few overlaps set aside by them are
// what makes the last corrected reads equal hifiasm's).  Two kernels:
//   k_snp_sites     one wavefront per 375-bp grid window: columns where at least two overlaps show a mismatch are candidate
//                   sites (cluster_advance :5585, markSNP_detail :4998); split_sub_list :5804 keeps a site when one alternative
//                   base dominates; every overlap covering a kept site leaves 0 (backbone's base), 1 (that base) or 2 (else) in
//                   the site's vector (InsertSNPVector, Correct.h:630).  Window cigars only: calculate_boundary_cigars :2310
//                   is not restated.
//   k_hap_partition one wavefront per read: generate_haplotypes_DP :6677 -- sites beside another site dropped, overlaps that are
//                   informative / not / informative again set aside (is_match 4), longest chains of mutually compatible sites
//                   enumerated (Preorder_Merge_Advance_Repeat :6233), a chain with support for both alleles
//                   (if_snp_vector_useful :6356) makes the overlaps with the other allele trans (is_match 2).
// oracle/asm.c:partition_read is the same algorithm, statement by statement.
#define FSV_SITE_WIN_CAP 255       // kept sites per grid window (their index in the window is a byte)
#define FSV_SITE_RAW_CAP 1024      // kept sites per read before the sites beside another site are dropped
#define FSV_SITE_READ_CAP 512      // ... and after (the chain DP's predecessor sets are 512-bit)
#define FSV_K7_GROUP_CAP 10000     // chains enumerated per read (the enumeration is exponential in ties; hifiasm has no bound)
struct SiteArgs {
    uint32_t *site_cnt;            // per grid window
    uint2 *site_rec;               // site records, handed out from a pool: {position in the read | homopolymer << 31, byte offset of the vector}
    uint32_t *site_off;            // per grid window: its first record
    uint32_t *rec_cursor; uint32_t rec_cap;
    int8_t *vec;                   // vector pool: one byte per overlap of the read, -1 = does not cover the site
    uint32_t *vec_cursor;          // bytes handed out
    uint32_t vec_cap;
    uint32_t *read_sites;          // per read: some window of it kept a site
    // the re-aligned junction cigars (k_bcig_tasks / k_bcig_accept); nullptr: the window cigars everywhere
    const int32_t *bc_idx;         // per window task: the junction task between it and the next window of its overlap, or -1
    const uint4 *bc_rec;           // per junction task: {bit 0: used, first column in x, columns | L << 16 | R << 24, -}
    const fsv_wpath *bc_paths;     // per junction task: its path
};

__device__ __forceinline__ void snp_sites_window(const ConsArgs &A, const uint32_t gw, const SiteArgs &S)
{
    __shared__ uint32_t s_cnt[FSV_WINDOW + 1][3];  // 16 bits each: A C | G T mismatch votes | x base without partner, -
    __shared__ int32_t s_cov[FSV_WINDOW + 2];
    __shared__ uint32_t s_path[64][27];
    __shared__ uint8_t s_alt[FSV_WINDOW + 1];      // 0, or 1 + the dominating other base of a kept site
    __shared__ uint8_t s_sidx[FSV_WINDOW + 1];     // kept site -> its index in the window
    __shared__ uint32_t s_cover, s_nsite, s_vbase, s_rbase;
    __shared__ uint32_t s_xraw[28];
    __shared__ uint32_t s_scan[64];
    const int lane = threadIdx.x;
    const uint4 gt = A.gwin_tab[gw];
    const uint32_t r = gt.x, pbase = gt.y, n_ovl = gt.z;
    const int g = (int)gt.w;
    const int xlen = A.read_len[r];
    const int gs = g * FSV_WINDOW, glen = min(FSV_WINDOW, xlen - gs);
    const uint32_t xw = A.word_off[r];
    const int xw0 = (gs >> 4) - 1;
    if (lane == 0) S.site_cnt[gw] = 0;
    if (lane < 28) { const int wi = xw0 + lane; s_xraw[lane] = (wi >= 0 && wi <= ((xlen + 15) >> 4)) ? A.store[xw + wi] : 0u; }
    for (int i = lane; i < (FSV_WINDOW + 1) * 3; i += 64) (&s_cnt[0][0])[i] = 0;
    for (int i = lane; i < FSV_WINDOW + 2; i += 64) s_cov[i] = 0;
    for (int i = lane; i < FSV_WINDOW + 1; i += 64) s_alt[i] = 0;
    if (lane == 0) { s_cover = 0; s_nsite = 0; }
    __syncthreads();
#define XB(p) ((s_xraw[((p) >> 4) - xw0] >> (((p) & 15) << 1)) & 3u)
#define CNT_ADD(c, b) atomicAdd(&s_cnt[(c)][(b) >> 1], 1u << (((b) & 1u) << 4))
#define CNT_GET(c, b) ((s_cnt[(c)][(b) >> 1] >> (((b) & 1u) << 4)) & 0xffffu)
    const uint32_t vstride = (n_ovl + 3u) & ~3u;
    for (int pass = 0; pass < 2; pass++) {
        for (uint32_t oi = lane; oi < n_ovl; oi += 64) {
            const uint4 oc = A.ovl_c[pbase + oi];
            const int o_x_s = (int)oc.x, o_n_win = (int)(oc.z & 0x7fffffffu);
            const int j = g - o_x_s / FSV_WINDOW;
            if (!(oc.z >> 31) || j < 0 || j >= o_n_win) continue;
            const uint32_t ti = oc.y + (uint32_t)j;
            const fsv_wpath *P = A.paths + ti;
            const uint4 h0 = *reinterpret_cast<const uint4 *>(P);
            if ((h0.w & 0xffu) != 1u) continue;
            const int xs = max(gs, o_x_s) - gs;
            if (pass == 0) atomicAdd(&s_cover, 1u);
            // What the overlap shows at the columns [lo, hi] (counted from the cigar's first column, which is column x0 of the read)
            // of one cigar.  pass 0: mismatches and x bases without a partner are tallied (markSNP_detail, Correct.cpp:4998);
            // pass 1: the kept sites get their evidence (addSNPtohaplotype_details :5247).  Returns the cigar's x columns.
            auto walk = [&](const fsv_wpath *Q, int x0, int lo, int hi) -> int {
                const uint4 q0 = *reinterpret_cast<const uint4 *>(Q);
                const int plen = (int)(int16_t)(q0.z & 0xffffu), ry0 = (int)q0.x, col0 = x0 - gs;
                if ((int16_t)(q0.z >> 16) == 0) {          // distance 0: all matches, the record carries no ops
                    if (pass == 1) for (int xi = max(lo, -col0); xi <= min(hi, plen - 1) && col0 + xi < glen; xi++) if (s_alt[col0 + xi]) S.vec[s_vbase + (uint32_t)s_sidx[col0 + xi] * vstride + oi] = 0;
                    return plen;
                }
                const uint2 q1 = *reinterpret_cast<const uint2 *>((const uint8_t *)Q + 16);
                const uint32_t y_word = q1.x; const int y_len = (int)q1.y, y_rev = (int)((q0.w >> 8) & 0xffu);
                const uint2 *src = reinterpret_cast<const uint2 *>(Q->ops);
#pragma unroll
                for (int i = 0; i < 13; i++) { const uint2 v = src[i]; s_path[lane][2 * i] = v.x; s_path[lane][2 * i + 1] = v.y; }
                int n2 = 0, n3 = 0;
                for (int p = 0; p < plen; p++) {
                    const uint32_t rest = s_path[lane][p >> 4] >> ((p & 15) << 1);
                    if (pass == 0 && rest == 0u) { p = (((p >> 4) + 1) << 4) - 1; continue; }    // the rest of the word: matches
                    const uint32_t op = rest & 3u;
                    if (op == 2u) { n2++; continue; }
                    const int xi = p - n2, c = col0 + xi;
                    if (xi >= lo && xi <= hi && c >= 0 && c < glen) {
                        if (pass == 0) {
                            if (op == 3u) CNT_ADD(c, 4u);
                            else if (op == 1u) CNT_ADD(c, fsv_base_at(A.store, y_word, y_len, y_rev, ry0 + p - n3));
                        } else {
                            const uint32_t alt = s_alt[c];
                            if (alt) {
                                int8_t v = 0;
                                if (op == 3u) v = 2;
                                else if (op == 1u) v = fsv_base_at(A.store, y_word, y_len, y_rev, ry0 + p - n3) + 1u == alt ? 1 : 2;
                                S.vec[s_vbase + (uint32_t)s_sidx[c] * vstride + oi] = v;
                            }
                        }
                    }
                    if (op == 3u) n3++;
                }
                return plen - n2;
            };
            // the window cigar in the middle; beside a junction whose re-aligned cigar is in use, that cigar (markSNP_advance :5054)
            int cur_beg = 0, cur_end = 0x7fffffff;
            if (S.bc_idx) {
                const int32_t sb = j >= 1 ? S.bc_idx[ti - 1] : -1, se = j + 1 < o_n_win ? S.bc_idx[ti] : -1;
                const uint4 rb = sb >= 0 ? S.bc_rec[sb] : make_uint4(0, 0, 0, 0), re = se >= 0 ? S.bc_rec[se] : make_uint4(0, 0, 0, 0);
                if ((rb.x | re.x) & 1u) {
                    const fsv_wtask t = A.tasks[ti];
                    const int x_total_start = t.x_start, x_length = t.x_len, x_total_end = x_total_start + x_length - 1;
                    if (rb.x & 1u) {
                        const int bx = (int)rb.y, blen = (int)(rb.z & 0xffffu), bL = (int)((rb.z >> 16) & 0xffu), bR = (int)(rb.z >> 24);
                        const int xleft = x_total_start - bx, xright = bx + blen - 1 - x_total_start + 1;
                        if (xleft > bL && xright > bR) { cur_beg = xright - bR; walk(S.bc_paths + sb, bx, xleft, xleft + (xright - bR) - 1); }
                    }
                    if (re.x & 1u) {
                        const int bx = (int)re.y, blen = (int)(re.z & 0xffffu), bL = (int)((re.z >> 16) & 0xffu), bR = (int)(re.z >> 24);
                        const int xleft = x_total_end - bx, xright = bx + blen - 1 - x_total_end + 1;
                        if (xleft > bL && xright > bR) { cur_end = (x_length - 1) - ((xleft + 1) - bL); walk(S.bc_paths + se, bx, bL, xleft); }
                    }
                }
            }
            const int xcols = walk(P, gs + xs, cur_beg, cur_end);
            if (pass == 0) {
                atomicAdd(&s_cov[xs], 1);
                atomicAdd(&s_cov[xs + xcols], -1);
            }
        }
        __syncthreads();
        if (pass == 1 || s_cover == 0u) break;
        // arrived[c] = prefix sum of the difference array; each lane owns the contiguous columns [c0, c1)
        const int per = (glen + 63) / 64, c0 = min(glen, lane * per), c1 = min(glen, c0 + per);
        int run = 0;
        for (int c = c0; c < c1; c++) run += s_cov[c];
        s_scan[lane] = (uint32_t)run;
        __syncthreads();
        int arrived = 0, mine = 0;
        for (int i = 0; i < lane; i++) arrived += (int)s_scan[i];
        for (int c = c0; c < c1; c++) {
            arrived += s_cov[c];
            // split_sub_list: occ_0 overlaps with the backbone's base, occ_1 with another base, occ_2 without a partner for it
            int oa[4], occ1 = 0, mx, mi = -1;
            const int occ2 = (int)CNT_GET(c, 4u);
#pragma unroll
            for (int b = 0; b < 4; b++) { oa[b] = (int)CNT_GET(c, (uint32_t)b); occ1 += oa[b]; }
            if (occ1 <= 1) continue;               // hap->flag > snp_threshold: at least two mismatches
            const int occ0 = arrived - occ1 - occ2, total = arrived;
            mx = occ2;
#pragma unroll
            for (int b = 0; b < 4; b++) if (oa[b] > mx) { mx = oa[b]; mi = b; }
            if (occ0 == 0 || mi < 0 || mx <= 1) continue;
            bool tie = false;
#pragma unroll
            for (int b = 0; b < 4; b++) if (oa[b] == mx && b != mi) tie = true;
            if (tie) continue;
            if ((double)(occ0 + 1 + mx) / (double)(total + 1) < 0.95) continue;
            if ((double)mx / (double)(total + 1 - (occ0 + 1)) < 0.70) continue;
            s_alt[c] = (uint8_t)(mi + 1);
            mine++;
        }
        s_scan[lane] = (uint32_t)mine;
        __syncthreads();
        int first = 0, all = 0;
        for (int i = 0; i < 64; i++) { if (i < lane) first += (int)s_scan[i]; all += (int)s_scan[i]; }
        if (all == 0) break;
        if (all > FSV_SITE_WIN_CAP) { if (lane == 0) atomicOr(&A.warn[r], (uint32_t)FSV_W_SITES); break; }
        if (lane == 0) {
            const uint32_t off = atomicAdd(S.vec_cursor, (uint32_t)all * vstride), roff = atomicAdd(S.rec_cursor, (uint32_t)all);
            s_vbase = off; s_rbase = roff;
            s_nsite = (off + (uint32_t)all * vstride <= S.vec_cap && roff + (uint32_t)all <= S.rec_cap) ? (uint32_t)all : 0u;
            if (!s_nsite) atomicOr(&A.warn[r], (uint32_t)FSV_W_SITES);
        }
        __syncthreads();
        if (!s_nsite) break;
        for (uint32_t i = lane; i < (uint32_t)all * vstride; i += 64) S.vec[s_vbase + i] = -1;
        for (int c = c0, k = first; c < c1; c++) {
            if (!s_alt[c]) continue;
            s_sidx[c] = (uint8_t)k;
            const int p = gs + c;
            const bool homo = homo_strict([&](int pp) { return XB(pp); }, p, xlen);
            S.site_rec[(size_t)s_rbase + k] = make_uint2((uint32_t)p | (homo ? 0x80000000u : 0u), s_vbase + (uint32_t)k * vstride);
            k++;
        }
        if (lane == 0) { S.site_cnt[gw] = (uint32_t)all; S.site_off[gw] = s_rbase; S.read_sites[r] = 1u; }
        __threadfence_block();
        __syncthreads();
    }
#undef XB
#undef CNT_ADD
#undef CNT_GET
}

// the windows k_consensus<., 1> marked (a fixed grid walks the list)
__global__ __launch_bounds__(64) void k_snp_sites(ConsArgs A, SiteArgs S, SiteLists L)
{
    const uint32_t n = L.win_n[0];
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        snp_sites_window(A, L.win_list[i], S);
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void k_hap_partition(ConsArgs A, SiteArgs S, fsv_ovl *__restrict__ ovl, uint4 *__restrict__ ovl_c, SiteLists L)
{
    __shared__ int32_t s_pos[FSV_SITE_RAW_CAP];
    __shared__ uint32_t s_voff[FSV_SITE_RAW_CAP];
    __shared__ uint16_t s_max[FSV_SITE_READ_CAP], s_order[FSV_SITE_READ_CAP];
    __shared__ uint32_t s_bt[FSV_SITE_READ_CAP][FSV_SITE_READ_CAP / 32];   // predecessors on a longest chain, as a bit set
    __shared__ uint16_t s_buf[FSV_SITE_READ_CAP], s_cur[FSV_SITE_READ_CAP]; // the chain being walked; per depth, the next predecessor to try
    __shared__ uint8_t s_visit[FSV_SITE_READ_CAP], s_keep[FSV_SITE_RAW_CAP];
    __shared__ uint32_t s_n;
    const int lane = threadIdx.x;
    __shared__ uint32_t s_redo;
    const uint32_t r = blockIdx.x;
    if (r >= A.n_reads || !S.read_sites[r]) return;
    const uint32_t g0 = A.gwin_off[r], g1 = A.gwin_off[r + 1];
    if (g0 == g1) return;
    if (lane == 0) s_redo = 0;
    const uint4 gt = A.gwin_tab[g0];
    const uint32_t pbase = gt.y, n_ovl = gt.z;
    // the read's kept sites in position order
    if (lane == 0) s_n = 0;
    __syncthreads();
    for (uint32_t gb = g0; gb < g1; gb += 64) {
        const uint32_t gw = gb + lane;
        const uint32_t c = gw < g1 ? S.site_cnt[gw] : 0u;
        uint32_t incl = c;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (lane >= d) incl += v; }
        const uint32_t base = s_n + incl - c;
        for (uint32_t k = 0; k < c; k++)
            if (base + k < FSV_SITE_RAW_CAP) { const uint2 rec = S.site_rec[(size_t)S.site_off[gw] + k]; s_pos[base + k] = (int32_t)rec.x; s_voff[base + k] = rec.y; }
        __syncthreads();
        if (lane == 63) s_n = base + c;
        __syncthreads();
    }
    int nS = (int)s_n;
    if (nS == 0) return;
    if (nS > FSV_SITE_RAW_CAP) { if (lane == 0) atomicOr(&A.warn[r], (uint32_t)FSV_W_SITES); return; }
    // a site directly beside another one is dropped
    if (nS > 1) {
        for (int j = lane; j < nS; j += 64) {
            const int p = s_pos[j] & 0x7fffffff;
            const bool left = j > 0 && p == (s_pos[j - 1] & 0x7fffffff) + 1, right = j + 1 < nS && p + 1 == (s_pos[j + 1] & 0x7fffffff);
            s_keep[j] = !(left || right);
        }
        __syncthreads();
        if (lane == 0) {
            int m = 0;
            for (int j = 0; j < nS; j++) if (s_keep[j]) { s_pos[m] = s_pos[j]; s_voff[m] = s_voff[j]; m++; }
            s_n = (uint32_t)m;
        }
        __syncthreads();
        nS = (int)s_n;
        if (nS == 0) return;
    }
    if (nS > FSV_SITE_READ_CAP) { if (lane == 0) atomicOr(&A.warn[r], (uint32_t)FSV_W_SITES); return; }
    // informative, not informative, informative again: the overlap is set aside
    for (uint32_t i = lane; i < n_ovl; i += 64) {
        if (!(ovl_c[pbase + i].z >> 31)) continue;
        int st = -1;
        for (int j = 0; j < nS; j++) {
            const int8_t v = S.vec[s_voff[j] + i];
            const bool inf = v == 0 || v == 1;
            if (st == -1) { if (inf) st = 0; }
            else if (st == 0) { if (!inf) st = 2; }
            else if (inf) { st = 3; break; }
        }
        if (st == 3) {
            for (int j = 0; j < nS; j++) S.vec[s_voff[j] + i] = 2;
            ovl[pbase + i].is_match = 4;
            ovl_c[pbase + i].z &= 0x7fffffffu;
            s_redo = 1u;
        }
    }
    __threadfence_block();
    __syncthreads();
    // longest chains of mutually compatible sites
    for (int i = 0; i < nS; i++) {
        if (lane < FSV_SITE_READ_CAP / 32) s_bt[i][lane] = 0;
        int best = 1;
        for (int j = 0; j < i; j++) {
            bool bad = false;
            for (uint32_t o = lane; o < n_ovl; o += 64) {
                const int8_t a = S.vec[s_voff[i] + o], b = S.vec[s_voff[j] + o];
                if (a != b && (a == 0 || a == 1) && (b == 0 || b == 1)) bad = true;
            }
            if (__ballot(bad)) continue;
            const int cand = (int)s_max[j] + 1;
            if (cand > best) {
                best = cand;
                if (lane < FSV_SITE_READ_CAP / 32) s_bt[i][lane] = 0;
            }
            if (cand == best && lane == 0) s_bt[i][j >> 5] |= 1u << (j & 31);
            __syncthreads();
        }
        if (lane == 0) { s_max[i] = (uint16_t)best; s_visit[i] = 0; }
        __syncthreads();
    }
    // longest first, ties in site order (a stable sort, as glibc's qsort is for arrays this small)
    for (int i = lane; i < nS; i += 64) {
        int rank = 0;
        for (int j = 0; j < nS; j++) rank += (s_max[j] > s_max[i]) || (s_max[j] == s_max[i] && j < i);
        s_order[rank] = (uint16_t)i;
    }
    __syncthreads();
    uint32_t n_groups = 0;
    for (int oi = 0; oi < nS; oi++) {
        const int root = s_order[oi];
        if (s_visit[root]) continue;          // uniform: s_visit is only written between barriers
        // depth-first over the predecessor sets, ascending site index (Preorder_Merge_Advance_Repeat)
        int depth = 0;
        if (lane == 0) { s_buf[0] = (uint16_t)root; s_cur[0] = 0; s_visit[root] = 1; }
        __syncthreads();
        while (depth >= 0) {
            const int id = s_buf[depth];
            bool leaf = true;
            for (int w = 0; w < FSV_SITE_READ_CAP / 32; w++) if (s_bt[id][w]) leaf = false;
            if (leaf) {
                if (n_groups <= FSV_K7_GROUP_CAP) {
                    // process_repeat_snps for the chain s_buf[0 .. depth]: first informative entry of every overlap
                    const int plen = depth + 1;
                    int occ0 = 0, occ1 = 0;
                    for (uint32_t ob = 0; ob < n_ovl; ob += 64) {
                        const uint32_t o = ob + lane;
                        int8_t rv = -1;
                        if (o < n_ovl) for (int k = 0; k < plen && rv == -1; k++) { const int8_t v = S.vec[s_voff[s_buf[k]] + o]; if (v == 0 || v == 1) rv = v; }
                        occ0 += __popcll(__ballot(rv == 0));
                        occ1 += __popcll(__ballot(rv == 1));
                    }
                    bool useful = false;
                    if (occ0 && occ1) {
                        const double low = (double)(occ0 + occ1) * 0.3;
                        if ((double)occ1 >= low && (double)occ0 >= low) useful = true;
                        else if (occ1 >= 5 && occ0 >= 5) useful = true;
                        else if (occ1 >= 3 && occ0 >= 3 && plen >= 2) {
                            int far = 0;
                            for (int k = 0; k < plen; k++) {
                                const int cur = s_pos[s_buf[k]] & 0x7fffffff;
                                bool nearby = false;
                                if (k > 0 && (s_pos[s_buf[k - 1]] & 0x7fffffff) - cur < 10) nearby = true;
                                if (k + 1 < plen && cur - (s_pos[s_buf[k + 1]] & 0x7fffffff) < 10) nearby = true;
                                if (!nearby) far++;
                            }
                            useful = far > 0;
                        }
                    }
                    if (useful)
                        for (uint32_t o = lane; o < n_ovl; o += 64) {
                            int8_t rv = -1;
                            for (int k = 0; k < plen && rv == -1; k++) { const int8_t v = S.vec[s_voff[s_buf[k]] + o]; if (v == 0 || v == 1) rv = v; }
                            if (rv == 1 && (ovl_c[pbase + o].z >> 31)) { ovl[pbase + o].is_match = 2; ovl_c[pbase + o].z &= 0x7fffffffu; s_redo = 1u; }
                        }
                    n_groups++;
                }
                depth--;
                continue;
            }
            // next predecessor of id at or after s_cur[depth]
            int nxt = -1;
            for (int j = s_cur[depth]; j < id; j++) if (s_bt[id][j >> 5] >> (j & 31) & 1u) { nxt = j; break; }
            __syncthreads();
            if (nxt < 0 || n_groups > FSV_K7_GROUP_CAP) { depth--; continue; }
            if (lane == 0) { s_cur[depth] = (uint16_t)(nxt + 1); s_buf[depth + 1] = (uint16_t)nxt; s_cur[depth + 1] = 0; s_visit[nxt] = 1; }
            __syncthreads();
            depth++;
        }
        __syncthreads();
    }
    // a read that lost an overlap: its windows get their consensus again
    if (s_redo) {
        uint32_t base = lane == 0 ? atomicAdd(&L.win_n[1], g1 - g0) : 0u;
        base = __shfl(base, 0);
        for (uint32_t i = lane; i < g1 - g0; i += 64) L.redo_list[base + i] = g0 + i;
    }
}

// ------------------------------------------------------------------------------------------------ k_newlen / k_repack
__global__ void k_newlen(const uint32_t *__restrict__ gwin_off, const uint16_t *__restrict__ cwin_len, uint32_t n_reads, int32_t *__restrict__ new_len,
                         uint32_t *__restrict__ lb = nullptr)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    int32_t L = 0;
    for (uint32_t g = gwin_off[r]; g < gwin_off[r + 1]; g++) { if (lb) lb[g] = (uint32_t)L; L += cwin_len[g]; }   // lb: where window g starts in the corrected read
    new_len[r] = L;
}

// One workgroup per read, a thread per output word: 16 bases gathered from the corrected windows of the read (optionally
// reverse-complemented).  The windows' starts in the corrected read (k_newlen's lb) are staged in LDS and searched there; with one
// thread per word of the whole store every thread searched the read table (15 dependent loads) and then walked its read's windows
// one dependent load at a time.
#define FSV_RP_WIN 1024
__global__ __launch_bounds__(256) void k_repack(const uint32_t *__restrict__ gwin_off, const uint32_t *__restrict__ lb,
                                                const uint8_t *__restrict__ cwin, const uint32_t *__restrict__ new_word_off,
                                                const int32_t *__restrict__ new_len, uint32_t n_reads, int rc,
                                                uint32_t *__restrict__ out, const uint32_t *__restrict__ only = nullptr)
{
    __shared__ int s_lb[FSV_RP_WIN + 1];
    const uint32_t r = blockIdx.x;
    if (r >= n_reads) return;
    if (only && !only[r]) return;      // the second pass only looks at reads some overlap deviates from
    const uint32_t g0 = gwin_off[r], nw = gwin_off[r + 1] - g0;
    const int len = new_len[r];
    const uint32_t w0 = new_word_off[r], slot = new_word_off[r + 1] - w0;
    const bool in_lds = nw <= FSV_RP_WIN;
    if (in_lds) {
        for (uint32_t i = threadIdx.x; i < nw; i += 256) s_lb[i] = (int)lb[g0 + i];
        if (threadIdx.x == 0) s_lb[nw] = len;
    }
    __syncthreads();
    auto LB = [&](uint32_t i) -> int { return in_lds ? s_lb[i] : (i < nw ? (int)lb[g0 + i] : len); };
    for (uint32_t w = threadIdx.x; w < slot; w += 256) {
        const int b0 = (int)w * 16;
        uint32_t v = 0;
        if (b0 < len && nw) {
            int src = rc ? len - 1 - b0 : b0;
            // the window holding the first source base: the last one that starts at or before it (empty windows share their start
            // with the next one), then step window by window
            uint32_t g = 0, hi = nw;
            while (hi - g > 1) { const uint32_t mid = (g + hi) >> 1; if (LB(mid) <= src) g = mid; else hi = mid; }
            for (int j = 0; j < 16 && b0 + j < len; j++) {
                uint32_t code = cwin[(size_t)(g0 + g) * FSV_CW_STRIDE + (src - LB(g))];
                if (rc) {
                    code = 3u - code;
                    src--;
                    while (g > 0 && src < LB(g)) g--;
                } else {
                    src++;
                    while (g + 1 < nw && src >= LB(g + 1)) g++;
                }
                v |= code << (2 * j);
            }
        }
        out[w0 + w] = v;
    }
}

// ------------------------------------------------------------------------------------------------ second consensus pass
// process_boundary (Correct.cpp:4453-4728) + merge_cigars (:4267): after the grid windows, every junction between two windows of a
// read once more.  Backbone = the 375 bases of the FIRST pass's result centred on the junction (read from a 2-bit copy of that
// result placed behind the round's read store, so K5 / K6 run on it as on any window task); every overlap that covers the start
// of the later window is re-aligned to it, threshold doubled once on failure; the inner bases (25 off either end) are replaced
// by the consensus of those alignments, from the first to the last column that keeps a base.  The replacement is handed to the
// two windows it touches as patches (k_bnd_apply) so that k_newlen / k_repack work on the windows as before.
// oracle/asm.c:correct_read (second_round) is the same, statement for statement.
#define FSV_BND_HALF 187
#define FSV_BND_SIDE 25
struct BndArgs {
    const fsv_wtask *tasks; const fsv_wpath *paths; const uint32_t *n_tasks;     // the round's window tasks and their paths
    const uint4 *ovl_c; const uint32_t *pair_base, *set_start; uint32_t n_sets; const uint32_t *pair_read;   // pair slot -> its query read
    const uint32_t *gwin_off, *lb; const uint16_t *cwin_len; const uint8_t *cov3; const uint32_t *read_dirty;
    const uint32_t *brel_off; uint32_t b_base;      // first-pass result of read r in the second-pass store: word b_base + brel_off[r]
    const uint8_t *thr_tab;
    fsv_wtask *tasks2; int32_t *idx2; uint32_t *n_tasks2;      // junction tasks; idx2[window task] = its junction task or -1
    uint32_t *bnd_flag, *bnd_list, *n_bnd;                    // per junction: bit 0 = has a task (and is in the list), the rest = alignments that match base for base
    const uint32_t *store2;                                   // the second-pass store: the round's reads, then the first pass's result
};

__global__ __launch_bounds__(256) void k_bnd_tasks(BndArgs A)
{
    const uint32_t n_tasks = min(*A.n_tasks, gridDim.x * blockDim.x);   // the grid covers the task bound
    uint32_t blk;
    if (!xcd_block((n_tasks + 255u) >> 8, blk)) return;
    const uint32_t ti = blk * blockDim.x + threadIdx.x;
    if (ti >= n_tasks) return;
    A.idx2[ti] = -1;
    const fsv_wtask t = A.tasks[ti];
    if (t.x_start % FSV_WINDOW != 0 || t.x_start == 0) return;    // only an overlap that covers the window's first base takes part
    // the tests are grouped so that the loads they need are in flight together: one test per load is one memory round trip per test
    const uint32_t p = t.ovl;
    const uint32_t r = A.pair_read[p];
    const uint4 oc = A.ovl_c[p];
    const uint4 h0 = *reinterpret_cast<const uint4 *>(A.paths + ti);
    const uint32_t dirty = A.read_dirty[r], gwo = A.gwin_off[r];
    // a clean read: every overlap matches it base for base, every junction alignment has distance 0
    if (!(dirty && (oc.z >> 31) && (h0.w & 0xffu) == 1u)) return;
    const uint32_t gw = gwo + (uint32_t)(t.x_start / FSV_WINDOW);
    const uint32_t cov = A.cov3[gw], cwl = A.cwin_len[gw];
    const int LB = (int)A.lb[gw];
    if (!cov || LB == 0) return;
    const int len_now = LB + (int)cwl;
    const int cws = max(0, LB - FSV_BND_HALF), cwe = min(len_now - 1, LB + FSV_BND_HALF - 1), blen = cwe - cws + 1;
    const int y_start = (int)h0.x - FSV_BND_HALF;
    if (y_start < 0 || blen < 1) return;
    // about half of the partner reads match the first pass's result base for base on the predicted diagonal (K5 would report distance 0
    // ending on that diagonal, K6 an all-match path): such an alignment only counts towards the junction's coverage
    const uint32_t xw2 = A.b_base + A.brel_off[r];
    if (y_start + blen <= t.y_len) {
        // (all six 64-base fetches of either read in flight: with an early exit per 16 bases an exact alignment -- the common case --
        // paid 24 dependent round trips)
        uint32_t acc = 0;
#pragma unroll
        for (int c = 0; c < 6; c++) {
            if (c * 64 < blen) {
                uint32_t xb[4], yb[4], yv[4];
                fetch64_x(A.store2, xw2, cws + c * 64, xb);
                fetch64(A.store2, t.y_word, t.y_len, t.y_rev, y_start + c * 64, yb, yv);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int b = c * 64 + j * 16;
                    if (b < blen) {
                        const int lim = min(16, blen - b);
                        acc |= (xb[j] ^ yb[j]) & (lim < 16 ? (1u << (2 * lim)) - 1u : 0xffffffffu);
                    }
                }
            }
        }
        const bool same = acc == 0u;
        if (same) { A.idx2[ti] = -2; atomicAdd(&A.bnd_flag[gw], 2u); return; }
    }
    const uint32_t slot = atomicAdd(A.n_tasks2, 1u);
    fsv_wtask w;
    w.x_word = xw2; w.y_word = t.y_word; w.x_start = cws; w.y_start = y_start; w.y_len = t.y_len;
    w.x_len = (uint16_t)blen; w.k = A.thr_tab[blen]; w.y_rev = t.y_rev; w.ovl = p; w.win = ti;
    A.tasks2[slot] = w;
    A.idx2[ti] = (int32_t)slot;
    if (!(atomicOr(&A.bnd_flag[gw], 1u) & 1u)) A.bnd_list[atomicAdd(A.n_bnd, 1u)] = gw;
}

// junction tasks K5 found no alignment for get the doubled threshold (Correct.cpp:4585-4626) and go round once more
__global__ __launch_bounds__(256) void k_bnd_retry(fsv_wtask *__restrict__ tasks2, const fsv_wres *__restrict__ res2, const uint32_t *__restrict__ n_tasks2,
                                                   fsv_wtask *__restrict__ tasks3, uint32_t *__restrict__ src3, uint32_t *__restrict__ n3, int k_cap)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_tasks2 || res2[i].err >= 0) return;
    fsv_wtask t = tasks2[i];
    t.k = (uint8_t)double_thr(t.k, t.x_len, k_cap);
    tasks2[i].k = t.k;
    const uint32_t j = atomicAdd(n3, 1u);
    tasks3[j] = t; src3[j] = i;
}

__global__ __launch_bounds__(256) void k_bnd_scatter(fsv_wres *__restrict__ res2, const fsv_wres *__restrict__ res3, const uint32_t *__restrict__ src3,
                                                     const uint32_t *__restrict__ n3)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < *n3) res2[src3[j]] = res3[j];
}

// what a junction hands to the two windows it touches: in window gw-1 the bases [ts, ts + told) become tnew bytes, in window
// gw the bases [hs, hs + hold) become hnew bytes (the tail bytes first in the junction's byte slot)
struct BndPatch { uint16_t ts, told, tnew, hs, hold, hnew, valid, pad; };

template <int EVC>
__global__ __launch_bounds__(64) void k_bnd_consensus(ConsArgs A, BndArgs B, const fsv_wpath *__restrict__ paths2, const uint32_t *__restrict__ store2,
                                                      BndPatch *__restrict__ patch, uint8_t *__restrict__ patch_bytes)
{
    __shared__ uint32_t s_cnt[FSV_WINDOW + 1][3];
    __shared__ int32_t s_cov[FSV_WINDOW + 2];
    __shared__ uint32_t s_path[64][27];
    __shared__ uint16_t s_evnext[EVC];
    __shared__ uint32_t s_evkey[EVC];
    __shared__ uint32_t s_evhead[FSV_WINDOW + 1];
    __shared__ uint16_t s_off[FSV_WINDOW + 2];      // where a column's output starts in the consensus
    __shared__ uint8_t s_own[FSV_WINDOW + 1];       // the column keeps a base (its own or another one)
    __shared__ uint32_t s_evn, s_cover, s_terr, s_nins, s_ndev;
    __shared__ uint16_t s_devlist[2 * FSV_BND_HALF + 2];   // columns some vote deviates at (as in consensus_window)
    __shared__ uint32_t s_fl[FSV_WINDOW + 1];      // votes of the mismatch runs that follow an insertion: 8 bits per base
    __shared__ uint16_t s_inslist[FSV_INSLIST];
    static_assert(sizeof(DagLds) <= sizeof(uint32_t) * 64 * 27, "the DAG scratch lives in the path buffer between the tally and the decisions");
    DagLds &s_dag = *reinterpret_cast<DagLds *>(&s_path[0][0]);
    __shared__ uint32_t s_xraw[28];
    const int lane = threadIdx.x;
    const uint32_t n_list = *B.n_bnd;
    for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
        __syncthreads();
        const uint32_t gw = B.bnd_list[li];
        const uint4 gt = A.gwin_tab[gw];
        const uint32_t r = gt.x, pbase = gt.y, n_ovl = gt.z;
        const int g = (int)gt.w;
        const int gs = g * FSV_WINDOW;
        const int LB = (int)B.lb[gw], len_now = LB + (int)B.cwin_len[gw];
        const int cws = max(0, LB - FSV_BND_HALF), cwe = min(len_now - 1, LB + FSV_BND_HALF - 1), blen = cwe - cws + 1;
        const uint32_t xw = B.b_base + B.brel_off[r];
        const int xw0 = (cws >> 4) - 1;
        if (lane == 0) patch[gw].valid = 0;
        if (lane < 28) { const int wi = xw0 + lane; s_xraw[lane] = (wi >= 0 && wi <= ((len_now + 15) >> 4)) ? store2[xw + wi] : 0u; }
        for (int i = lane; i < (FSV_WINDOW + 1) * 3; i += 64) (&s_cnt[0][0])[i] = 0;
        for (int i = lane; i < FSV_WINDOW + 2; i += 64) s_cov[i] = 0;
        for (int i = lane; i < FSV_WINDOW + 1; i += 64) s_evhead[i] = 0xffffu;
        for (int i = lane; i < FSV_WINDOW + 1; i += 64) s_fl[i] = 0;
        if (lane == 0) { s_evn = 0; s_cover = 0; s_terr = 0; s_nins = 0; s_ndev = 0; }
        __syncthreads();
        const ConsLds CL = {s_cnt, s_cov, s_fl, &s_evn, s_evkey, s_evnext, s_evhead};
#define XB(p) ((s_xraw[((p) >> 4) - xw0] >> (((p) & 15) << 1)) & 3u)
#define CNT_ADD(c, b) atomicAdd(&s_cnt[(c)][(b) >> 1], 1u << (((b) & 1u) << 4))
#define CNT_GET(c, b) ((s_cnt[(c)][(b) >> 1] >> (((b) & 1u) << 4)) & 0xffffu)
        for (uint32_t oi = lane; oi < n_ovl; oi += 64) {
            const uint4 oc = A.ovl_c[pbase + oi];
            const int o_x_s = (int)oc.x, o_n_win = (int)(oc.z & 0x7fffffffu);
            const int j = g - o_x_s / FSV_WINDOW;
            if (!(oc.z >> 31) || j < 0 || j >= o_n_win || o_x_s > gs) continue;
            const int32_t slot = B.idx2[oc.y + (uint32_t)j];
            if (slot == -2) { atomicAdd(&s_cover, 1u); atomicAdd(&s_cov[0], 1); atomicAdd(&s_cov[blen], -1); continue; }   // matches base for base
            if (slot < 0) continue;
            const fsv_wpath *P = paths2 + slot;
            const uint4 h0 = *reinterpret_cast<const uint4 *>(P);
            if ((h0.w & 0xffu) != 1u) continue;
            const uint2 h1 = *reinterpret_cast<const uint2 *>((const uint8_t *)P + 16);
            const int perr = (int)(int16_t)(h0.z >> 16);
            atomicAdd(&s_cover, 1u);
            const int ry_start = (int)h0.x, plen = (int)(int16_t)(h0.z & 0xffffu);
            int n2 = 0;
            if (perr != 0) {
                atomicAdd(&s_terr, (uint32_t)perr);
                const uint2 *src = reinterpret_cast<const uint2 *>(P->ops);
                uint32_t nz = 0;
#pragma unroll
                for (int i = 0; i < 13; i++) {
                    const uint2 v = src[i];
                    s_path[lane][2 * i] = v.x; s_path[lane][2 * i + 1] = v.y;
                    nz |= (v.x ? 1u << (2 * i) : 0u) | (v.y ? 2u << (2 * i) : 0u);
                }
                n2 = cons_walk<EVC>(CL, s_path[lane], nz, plen, 0, blen, false, A.store, h1.x, (int)h1.y, (int)((h0.w >> 8) & 0xffu), ry_start);
            }
            atomicAdd(&s_cov[0], 1);
            atomicAdd(&s_cov[plen - n2], -1);
        }
        __syncthreads();
        if (s_cover < 3u || s_terr == 0u) continue;       // MIN_COVERAGE_THRESHOLD; "if there are no error, we do not need correction"
        if (s_evn > (uint32_t)EVC && lane == 0) atomicOr(&A.warn[r], (uint32_t)FSV_W_INS_EVENTS);
        const int per = (blen + 63) / 64, c0 = min(blen, lane * per), c1 = min(blen, c0 + per);
        int run = 0, frun = 0;
        for (int c = c0; c < c1; c++) {
            const int v = s_cov[c];
            run += COV_LO(v); frun += COV_HI(v);
            if (CNT_GET(c, 5u)) {
                const uint32_t head = s_evhead[c];
                bool same = true;
                if (head != 0xffffu) { const uint32_t k0 = s_evkey[head]; for (uint32_t i = s_evnext[head]; i != 0xffffu; i = s_evnext[i]) if (s_evkey[i] != k0) { same = false; break; } }
                if (!same && A.ins_dag) { const uint32_t k = atomicAdd(&s_nins, 1u); if (k < FSV_INSLIST) s_inslist[k] = (uint16_t)c; }
            }
        }
        int arrived = wave_incl_sum(run, lane) - run, farrived = wave_incl_sum(frun, lane) - frun;
        __syncthreads();
        if (lane == 0 && s_nins) answer_insertions<EVC>(s_dag, s_evhead, s_evkey, s_evnext, s_inslist, (int)s_nins, blen);
        __syncthreads();
        uint8_t (*s_out)[14] = reinterpret_cast<uint8_t (*)[14]>(&s_path[0][0]);
        bool differs = false;
        // columns without a deviating vote keep the backbone's base; the others are listed and decided a lane each (consensus_window)
        for (int c = c0; c < c1; c++) {
            arrived += COV_LO(s_cov[c]);
            farrived += COV_HI(s_cov[c]);
            s_out[c][0] = 1; s_out[c][1] = (uint8_t)XB(cws + c);
            s_own[c] = 1;
            if ((s_cnt[c][0] | s_cnt[c][1] | s_cnt[c][2]) != 0u) {
                s_cov[c] = (int32_t)(((uint32_t)arrived & 0xffffu) | ((uint32_t)farrived << 16));
                s_devlist[atomicAdd(&s_ndev, 1u)] = (uint16_t)c;
            }
        }
        __syncthreads();
        for (uint32_t e = lane; e < s_ndev; e += 64) {
            const int c = (int)s_devlist[e];
            arrived = (int)(int16_t)((uint32_t)s_cov[c] & 0xffffu); farrived = (int)(int16_t)((uint32_t)s_cov[c] >> 16);
            const int p = cws + c;
            const uint32_t own = XB(p);
            const bool homo = c > 0 && homo_strict([&](int pp) { return XB(pp); }, p - 1, len_now);
            int W[4], Ifl[4], dev = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) { W[b] = (int)CNT_GET(c, (uint32_t)b); dev += W[b]; Ifl[b] = (int)((s_fl[c] >> (b << 3)) & 0xffu); }
            const int dl = (int)CNT_GET(c, 4u), ni = (int)CNT_GET(c, 5u);
#pragma unroll
            for (int b = 0; b < 4; b++) if ((int)own == b) { W[b] += arrived - dev - dl + 1; Ifl[b] = farrived; }
            int mi = 0; uint32_t ikey = 0;
            if (ni) {
                const uint32_t hv = s_evhead[c], head = hv & 0xffffu;
                if (head != 0xffffu) {
                    ikey = s_evkey[head];
                    if (hv & 0x10000u) mi = (int)s_evnext[head];
                    else if (A.ins_dag) { for (uint32_t i = head; i != 0xffffu; i = s_evnext[i]) mi++; }
                    else most_frequent_insertion(s_evkey, s_evnext, head, mi, ikey);
                }
            }
            const bool kept = poa_decide(W, Ifl, dl, ni, mi, ikey, (int)own, homo, &s_out[c][0]);
            s_own[c] = kept;
            if (s_out[c][0] != 1 || s_out[c][1] != (uint8_t)own) differs = true;
        }
        __syncthreads();
        int mine = 0;
        for (int c = c0; c < c1; c++) mine += s_out[c][0];
        if (__ballot(differs) == 0ull) continue;          // the new cigar is one run of matches
        int off = wave_incl_sum(mine, lane) - mine;
        for (int c = c0; c < c1; c++) { s_off[c] = (uint16_t)off; off += s_out[c][0]; }
        // the first and the last column to replace: the first kept column at or after 25 / at or after blen - 1 - 25
        const int sb = FSV_BND_SIDE, eb = blen - 1 - FSV_BND_SIDE;
        int fs = 0x7fffffff, fe = 0x7fffffff;
        for (int c = c0; c < c1; c++) { if (s_own[c] && c >= sb && c < fs) fs = c; if (s_own[c] && c >= eb && c < fe) fe = c; }
        for (int d = 32; d >= 1; d >>= 1) { fs = min(fs, __shfl_xor(fs, d)); fe = min(fe, __shfl_xor(fe, d)); }
        __syncthreads();
        if (eb <= sb || fs == 0x7fffffff || fe == 0x7fffffff) continue;   // "if there are some gap at the end of x, it very likely miscorrection"
        const int o0 = (int)s_off[fs] + s_out[fs][0] - 1, o1 = (int)s_off[fe] + s_out[fe][0] - 1;   // the two kept bases in the consensus
        const int R0 = cws + fs, R1 = cws + fe, sc = LB - cws;    // first-pass coordinates of the stretch; sc: the later window's first column
        const int o_split = sc <= fs ? o0 : (sc > fe ? o1 + 1 : (int)s_off[sc]);
        const int lb_prev = (int)B.lb[gw - 1];
        BndPatch bp;
        bp.valid = 1; bp.pad = 0;
        bp.ts = (uint16_t)max(0, R0 - lb_prev); bp.told = (uint16_t)max(0, min(R1, LB - 1) - R0 + 1); bp.tnew = (uint16_t)(o_split - o0);
        bp.hs = (uint16_t)(max(R0, LB) - LB); bp.hold = (uint16_t)max(0, R1 - max(R0, LB) + 1); bp.hnew = (uint16_t)(o1 + 1 - o_split);
        if (R0 < lb_prev || o1 + 1 - o0 > FSV_CW_STRIDE) {     // the stretch would reach a third window / outgrow its slot: left as the first pass had it
            if (lane == 0) atomicOr(&A.warn[r], (uint32_t)FSV_W_WINDOW_KEPT);
            continue;
        }
        uint8_t *dst = patch_bytes + (size_t)gw * FSV_CW_STRIDE;
        for (int c = c0; c < c1; c++)
            for (int b = 0; b < s_out[c][0]; b++) { const int pos = (int)s_off[c] + b; if (pos >= o0 && pos <= o1) dst[pos - o0] = s_out[c][1 + b]; }
        if (lane == 0) patch[gw] = bp;
#undef XB
#undef CNT_ADD
#undef CNT_GET
    }
}

// one wavefront per grid window: the window with the patches of its two junctions applied, in place
__global__ __launch_bounds__(64) void k_bnd_apply(const uint32_t *__restrict__ gwin_read, const uint32_t *__restrict__ gwin_off, const BndPatch *__restrict__ patch,
                                                  const uint8_t *__restrict__ patch_bytes, const uint32_t *__restrict__ bnd_flag, uint32_t n_gwin,
                                                  uint8_t *__restrict__ cwin, uint16_t *__restrict__ cwin_len, uint32_t *__restrict__ changed, uint32_t *__restrict__ warn)
{
    __shared__ uint8_t s_old[FSV_CW_STRIDE];
    const uint32_t gw = blockIdx.x;
    if (gw >= n_gwin) return;
    const uint32_t r = gwin_read[gw];
    const bool has_h = gw > gwin_off[r] && (bnd_flag[gw] & 1u) && patch[gw].valid;
    const bool has_t = gw + 1 < gwin_off[r + 1] && (bnd_flag[gw + 1] & 1u) && patch[gw + 1].valid;
    if (!has_h && !has_t) return;
    const int lane = threadIdx.x;
    const int len = cwin_len[gw];
    uint8_t *w = cwin + (size_t)gw * FSV_CW_STRIDE;
    for (int i = lane; i < len; i += 64) s_old[i] = w[i];
    __syncthreads();
    int hs = 0, hold = 0, hnew = 0, ts = len, told = 0, tnew = 0;
    const uint8_t *hb = nullptr, *tb = nullptr;
    if (has_h) { const BndPatch p = patch[gw]; hs = p.hs; hold = p.hold; hnew = p.hnew; hb = patch_bytes + (size_t)gw * FSV_CW_STRIDE + p.tnew; }
    if (has_t) { const BndPatch p = patch[gw + 1]; ts = p.ts; told = p.told; tnew = p.tnew; tb = patch_bytes + (size_t)(gw + 1) * FSV_CW_STRIDE; }
    if (told == 0 && tnew == 0) ts = len;
    const int new_len = len - hold + hnew - told + tnew;
    if (hs + hold > ts || ts + told > len || new_len > FSV_CW_STRIDE || new_len < 0) {   // cannot happen with windows of ~375 bases: keep the first pass
        if (lane == 0) atomicOr(&warn[r], (uint32_t)FSV_W_WINDOW_KEPT);
        return;
    }
    // [0, hs) | head patch | [hs + hold, ts) | tail patch | [ts + told, len)
    const int a1 = hs, a2 = a1 + hnew, a3 = a2 + (ts - hs - hold), a4 = a3 + tnew;
    for (int i = lane; i < new_len; i += 64) {
        uint8_t v;
        if (i < a1) v = s_old[i];
        else if (i < a2) v = hb[i - a1];
        else if (i < a3) v = s_old[hs + hold + (i - a2)];
        else if (i < a4) v = tb[i - a3];
        else v = s_old[ts + told + (i - a4)];
        w[i] = v;
    }
    if (lane == 0) { cwin_len[gw] = (uint16_t)new_len; changed[r] = 1u; }
}

// ------------------------------------------------------------------------------------------------ k_exact
// if_exact_match (Assembly.cpp:894-974): the two overlap intervals must be the same string.  One wavefront per unordered
// pair: the overlap of t on q covers the same bases as the overlap of q on t, so one comparison decides both slots.
// The comparison issues four 16-base fetches per lane before it looks at any of them (the early exit costs a memory
// round trip per test, and most valid overlaps of the final pass are exact).
// An exact overlap as the host layout reads it (32 B; q, t are read indices inside the set).
struct fsv_hit { uint32_t q, t; int32_t x_s, x_e, y_s, y_e; uint32_t rev, slot; };   // slot: the overlap slot; bit 31 set = exact

__global__ __launch_bounds__(64) void k_exact(const uint32_t *__restrict__ store, const uint32_t *__restrict__ word_off,
                                              const int32_t *__restrict__ read_len, const uint32_t *__restrict__ read_set,
                                              const uint32_t *__restrict__ pair_base, const uint4 *__restrict__ upair_tab,
                                              const fsv_ovl *__restrict__ ovl, fsv_hit *__restrict__ hits, uint32_t *__restrict__ set_hits,
                                              uint8_t *__restrict__ exact_flag)
{
    const uint4 pt = upair_tab[blockIdx.x];
    const int lane = threadIdx.x;
    fsv_ovl o = ovl[pt.z];
    if (lane == 0) exact_flag[blockIdx.x] = 0;
    if (!o.valid) return;
    const uint32_t rq = pt.x + o.q, rt = pt.x + o.t;
    const int L = o.x_e - o.x_s + 1;
    bool same = (L == o.y_e - o.y_s + 1);
    if (same) {
        const uint32_t xw = word_off[rq], yw = word_off[rt];
        const int ylen = read_len[rt];
        uint32_t acc = 0;
        for (int i0 = lane * 16; i0 < L; i0 += 4 * 64 * 16) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * 64 * 16;
                if (i < L) {
                    const uint32_t xb = fetch16_x(store, xw, o.x_s + i);
                    const Bases16 yb = fetch16(store, yw, ylen, o.rev, o.y_s + i);
                    uint32_t d = xb ^ yb.bits;
                    const int lim = min(16, L - i);
                    if (lim < 16) d &= (1u << (2 * lim)) - 1u;
                    acc |= d;
                }
            }
            if (__any(acc != 0u)) break;
        }
        same = !__any(acc != 0u);
    }
    if (!same) return;
    if (lane == 0) exact_flag[blockIdx.x] = 1;
    // exact hits are gathered per set for the host layout (a few per cent of the slots): the set's hit segment starts at its
    // first ordered-pair slot and a per-set counter hands out places -- one atomic per pair, spread over the sets' addresses
    const uint32_t s = read_set[pt.x];
    uint32_t at = 0;
    if (lane == 0) at = atomicAdd(&set_hits[s], 2u);
    at = __shfl(at, 0, 64);
    if (lane < 2) {
        const uint32_t slot = lane ? pt.w : pt.z;
        if (lane) o = ovl[slot];
        fsv_hit h; h.q = o.q; h.t = o.t; h.x_s = o.x_s; h.x_e = o.x_e; h.y_s = o.y_s; h.y_e = o.y_e; h.rev = o.rev; h.slot = slot | 0x80000000u;   // bit 31: an exact overlap (hifiasm's el)
        hits[pair_base[s] + at + lane] = h;
    }
}

// the per-set hit segments, packed back to back for one D2H copy: one block per set, 8 words per record
__global__ __launch_bounds__(256) void k_hits_compact(const fsv_hit *__restrict__ hits, const uint32_t *__restrict__ pair_base,
                                                      const uint32_t *__restrict__ hit_first, fsv_hit *__restrict__ out)
{
    const uint32_t s = blockIdx.x;
    const uint32_t n = (hit_first[s + 1] - hit_first[s]) * 8u;
    const uint32_t *src = (const uint32_t *)(hits + pair_base[s]);
    uint32_t *dst = (uint32_t *)(out + hit_first[s]);
    for (uint32_t i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

// Inexact overlaps for the layout (update_overlaps, Assembly.cpp:975-1083 as called by worker_ov_final :1284-1306): a pair
// without an exact overlap whose overlap the last correction round verified is chained again with hifiasm's final bandwidth
// (0.001, gapped coordinates) and accepted per direction when strand and coordinates agree with the verified overlap (both
// ends of either read within 10 % of the longer span).
__global__ void k_inexact_list(const uint4 *__restrict__ upair_tab, const uint8_t *__restrict__ exact_flag, const fsv_ovl *__restrict__ prev,
                               uint32_t n_upairs, uint32_t *__restrict__ list, uint32_t *__restrict__ n_list)
{
    const uint32_t up = blockIdx.x * blockDim.x + threadIdx.x;
    if (up >= n_upairs || exact_flag[up]) return;
    const uint4 pt = upair_tab[up];
    const fsv_ovl a = prev[pt.z], b = prev[pt.w];
    if ((a.valid && a.is_match == 1) || (b.valid && b.is_match == 1)) list[atomicAdd(n_list, 1u)] = up;
}

__global__ void k_accept_inexact(const uint4 *__restrict__ upair_tab, const uint32_t *__restrict__ list, uint32_t n_list,
                                 const fsv_ovl *__restrict__ ovl, const fsv_ovl *__restrict__ prev, const uint32_t *__restrict__ read_set,
                                 const uint32_t *__restrict__ pair_base, fsv_hit *__restrict__ hits, uint32_t *__restrict__ set_hits,
                                 const uint32_t *__restrict__ n_dev = nullptr)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n_list = *n_dev;
    if (i >= 2 * n_list) return;
    const uint4 pt = upair_tab[list[i >> 1]];
    const uint32_t slot = (i & 1) ? pt.w : pt.z;
    const fsv_ovl o = ovl[slot], pv = prev[slot];
    if (!o.valid || !pv.valid || pv.is_match != 1 || pv.rev != o.rev) return;
    const int lx = pv.x_e - pv.x_s + 1, ly = pv.y_e - pv.y_s + 1, L = max(lx, ly) / 10;
    const bool ok = (abs(o.x_s - pv.x_s) < L && abs(o.x_e - pv.x_e) < L) || (abs(o.y_s - pv.y_s) < L && abs(o.y_e - pv.y_e) < L);
    if (!ok) return;
    const uint32_t s = read_set[pt.x];
    fsv_hit h; h.q = o.q; h.t = o.t; h.x_s = o.x_s; h.x_e = o.x_e; h.y_s = o.y_s; h.y_e = o.y_e; h.rev = o.rev; h.slot = slot;
    hits[pair_base[s] + atomicAdd(&set_hits[s], 1u)] = h;
}

// ------------------------------------------------------------------------------------------------ k_stitch
struct fsv_piece { uint32_t read; uint32_t rev; uint32_t len; uint32_t pad; uint64_t dst; };
__global__ __launch_bounds__(256) void k_stitch(const uint32_t *__restrict__ store, const uint32_t *__restrict__ word_off,
                                                const int32_t *__restrict__ read_len, const fsv_piece *__restrict__ pieces, char *__restrict__ out)
{
    const fsv_piece pc = pieces[blockIdx.x];
    const uint32_t w = word_off[pc.read];
    const int len = read_len[pc.read];
    for (uint32_t i = threadIdx.x; i < pc.len; i += blockDim.x) out[pc.dst + i] = "ACGT"[fsv_base_at(store, w, len, (int)pc.rev, (int)i)];
}

__global__ void k_unpack_reads(const uint32_t *__restrict__ store, const uint32_t *__restrict__ word_off, const int32_t *__restrict__ read_len,
                               const uint64_t *__restrict__ dst_off, char *__restrict__ out)
{
    const uint32_t r = blockIdx.x;
    const uint32_t w = word_off[r];
    const int len = read_len[r];
    for (int i = threadIdx.x; i < len; i += blockDim.x) out[dst_off[r] + i] = "ACGT"[fsv_base_fwd(store, w, i)];
}

} // namespace

namespace {
// ------------------------------------------------------------------------------------------------ k_sketch_fast
// Position-parallel form of ha_sketch (sketch.cpp:39-137) for odd k (hifiasm's 51, minimap2's 19): no sequential replay.
//   phase 0  homopolymer compression in parallel: run ends are found per 16-base word, a block scan gives every kept
//            base its entry index; the compressed bases go to two bit planes, the run-end positions to an array
//            (per-read slices of an HBM scratch, L2-resident for the block that wrote them);
//   phase 1  every entry's k-mer is cut out of the bit planes with funnel shifts (forward strand = bit-reversed,
//            reverse strand = complemented), hashed, and the span is a difference of two run-end positions;
//   phase 2  ha_sketch reports an entry iff it equals the minimum of some window of w entries that ends at or after the
//            first full one (as the window's "best" or as an identical-k-mer copy); window minima and the test are
//            brute-force scans of an LDS tile.  The irregular first full window (l == w+k-1: copies of the previous
//            partial window's minimum are reported, that minimum itself only if the incoming k-mer is larger) and reads
//            shorter than one window (only the last minimum) are handled explicitly.
// Equivalent to the monotone-deque replay in k_sketch (which stays for even k); both are checked against the oracle.
#define SKF_T 1024
#define SKF_V ((SKF_T + 2 * 256 + 255) / 256)   // elements of the doubling passes per thread
__global__ __launch_bounds__(256) void k_sketch_fast(const uint32_t *__restrict__ store, const uint32_t *__restrict__ word_off,
                                                     const int32_t *__restrict__ read_len, const uint32_t *__restrict__ mz_off,
                                                     fsv_mz *__restrict__ mz, uint32_t *__restrict__ mz_cnt, uint32_t n_reads, int w, int k,
                                                     int hpc, uint32_t *__restrict__ warn, const uint8_t *__restrict__ w_per_read,
                                                     uint32_t *__restrict__ sc_ends, uint32_t *__restrict__ sc_low, uint32_t *__restrict__ sc_high,
                                                     const uint32_t *__restrict__ only_changed)
{
    __shared__ uint64_t s_h[SKF_T + 2 * 256];   // w <= 255
    __shared__ uint64_t s_wmin[SKF_T + 2 * 256];
    __shared__ uint32_t s_scan[4];
    __shared__ uint32_t s_carry;
    __shared__ uint64_t s_am, s_ah;   // start anomaly: minimum of the partial window, hash of entry T0
    __shared__ int s_abest, s_short;  // its rightmost position; the single minimizer of a read shorter than one window
    const int tid = threadIdx.x;
    const uint32_t r = blockIdx.x;
    if (r >= n_reads) return;
    // a read the last correction round left as it was keeps the minimizers of that round (same sequence, same slot)
    if (only_changed && !only_changed[r]) return;
    const uint32_t woff = word_off[r];
    const int len = read_len[r];
    const uint32_t cap = mz_off[r + 1] - mz_off[r];
    fsv_mz *out = mz + mz_off[r];
    if (w_per_read) w = w_per_read[r];
    if (tid == 0) mz_cnt[r] = 0;   // (the first emit comes after several barriers)
    uint32_t *ends = sc_ends + (size_t)woff * 16;          // entry -> index of the run's last base
    uint32_t *low = sc_low + woff + r, *high = sc_high + woff + r; // bit planes of the compressed bases (zeroed by the host)
    const uint64_t NONE = ~0ull;
    const uint64_t kmask = (1ull << k) - 1;
    // ---- phase 0
    const int nwords = (len + 15) >> 4;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int wbase = 0; wbase < nwords; wbase += 256) {
        const int wi = wbase + tid;
        uint32_t flags = 0, word = 0;
        int nb = 0;
        if (wi < nwords) {
            word = store[woff + wi];
            nb = min(16, len - wi * 16);
            if (hpc) {
                const uint32_t nextb = (wi + 1 < nwords) ? (store[woff + wi + 1] & 3u) : 4u;
                // base j ends a run when it differs from base j+1 (the read's last base always does)
                const uint32_t shifted = (word >> 2) | (nextb << 30);
                uint32_t d = word ^ shifted;
                d = (d | (d >> 1)) & 0x55555555u; // field j non-zero <=> base j != base j+1
                // even bits -> 16-bit mask; the read's last base always ends a run
                d = (d | (d >> 1)) & 0x33333333u; d = (d | (d >> 2)) & 0x0f0f0f0fu; d = (d | (d >> 4)) & 0x00ff00ffu; d = (d | (d >> 8)) & 0xffffu;
                flags = nb >= 16 ? d : (d & ((1u << nb) - 1u));
                if (len - 1 - wi * 16 < 16) flags |= 1u << (len - 1 - wi * 16);
            } else flags = nb >= 16 ? 0xffffu : ((1u << nb) - 1u);
        }
        const uint32_t cnt = __popc(flags);
        // exclusive scan over the 256 threads: shuffle scan inside each wave, the four wave totals through LDS
        uint32_t incl = cnt;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o2 = __shfl_up(incl, off, 64); if ((tid & 63) >= off) incl += o2; }
        if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
        __syncthreads();
        uint32_t wave_before = 0, tile_total = 0;
#pragma unroll
        for (int wv = 0; wv < 4; wv++) { const uint32_t v = s_scan[wv]; if (wv < (tid >> 6)) wave_before += v; tile_total += v; }
        const uint32_t base = s_carry + wave_before + incl - cnt;
        if (cnt) {
            uint32_t lo = 0, hi = 0, rank = 0;
            for (int j = 0; j < nb; j++)
                if ((flags >> j) & 1u) {
                    const uint32_t b = (word >> (2 * j)) & 3u;
                    lo |= (b & 1u) << rank; hi |= (b >> 1) << rank;
                    ends[base + rank] = (uint32_t)(wi * 16 + j);
                    rank++;
                }
            const uint32_t wd = base >> 5, sh = base & 31u;
            atomicOr(&low[wd], lo << sh); atomicOr(&high[wd], hi << sh);
            if (sh + cnt > 32) { atomicOr(&low[wd + 1], lo >> (32 - sh)); atomicOr(&high[wd + 1], hi >> (32 - sh)); }
        }
        __syncthreads();
        if (tid == 0) s_carry += tile_total;
        __syncthreads();
    }
    const int M = (int)s_carry; // entries
    const int T0 = w + k - 2;   // entry index of the first full window (l == w+k-1)
    __threadfence_block();
    __syncthreads();
    // k consecutive plane bits starting at entry a (a >= 0), bit i = entry a+i
    auto cut = [&](const uint32_t *pl, int a) -> uint64_t {
        const int wd = a >> 5, sh = a & 31;
        const uint64_t lo64 = (uint64_t)pl[wd] | (uint64_t)pl[wd + 1] << 32;
        uint64_t v = lo64 >> sh;
        if (sh) v |= (uint64_t)pl[wd + 2] << (64 - sh);
        return v & kmask;
    };
    auto entry_hash = [&](int e, int *z_out) -> uint64_t {
        if (e < k - 1 || e >= M) return NONE;
        const int span = (int)ends[e] - (e - k >= 0 ? (int)ends[e - k] : -1);
        if (hpc && span >= 256) return NONE;
        const uint64_t lo = cut(low, e - k + 1), hi = cut(high, e - k + 1);
        // forward strand: oldest base in the top bit; reverse strand: complement, oldest base in bit 0
        const uint64_t f0 = __brevll(lo) >> (64 - k), f1 = __brevll(hi) >> (64 - k);
        const uint64_t r0 = ~lo & kmask, r1 = ~hi & kmask;
        const int z = f1 < r1 ? 0 : 1;
        if (z_out) *z_out = z;
        return mix64(z ? r0 : f0) + mix64(z ? r1 : f1);   // the strand is chosen first: two hashes per entry, not four
    };
    auto emit = [&](int p) {
        int z = 0;
        const uint64_t h = entry_hash(p, &z);
        const int span = hpc ? (int)ends[p] - (p - k >= 0 ? (int)ends[p - k] : -1) : k;
        const uint32_t at = atomicAdd(&mz_cnt[r], 1u);
        if (at < cap) { fsv_mz m; m.hash = h; m.pos = ends[p]; m.rev = (uint8_t)z; m.span = (uint8_t)span; m.pad = 0; out[at] = m; }
        else atomicOr(&warn[r], (uint32_t)FSV_W_MZ_TRUNC);
    };
    for (int t0 = 0; t0 < M; t0 += SKF_T) {
        const int e0 = t0 - (w - 1); // entry held by s_h[0]
        for (int idx = tid; idx < SKF_T + 2 * (w - 1); idx += 256) s_h[idx] = entry_hash(e0 + idx, nullptr);
        __syncthreads();
        // window minima: s_wmin[i] = min over entries (t0+i)-(w-1) .. (t0+i) = min(s_h[i .. i+w-1]), by doubling: after the
        // pass with distance d an element covers 2d entries; two overlapping power-of-two ranges make the window of w
        const int NW = SKF_T + 2 * (w - 1);
        int p2 = 1;
        while (p2 * 2 <= w) p2 <<= 1;
        // every thread keeps its own elements in registers between the passes and only reads its partner's from the tile
        uint64_t v[SKF_V];
#pragma unroll
        for (int c = 0; c < SKF_V; c++) { const int i = tid + 256 * c; v[c] = i < NW ? s_h[i] : NONE; }
        auto pass = [&](const uint64_t *src, int d, bool is_max) {   // s_wmin[i] = op(own[i], src[i+d]) for every i < NW
#pragma unroll
            for (int c = 0; c < SKF_V; c++) {
                const int i = tid + 256 * c;
                if (i < NW) {
                    const uint64_t b = i + d < NW ? src[i + d] : (is_max ? 0ull : NONE);
                    v[c] = is_max ? max(v[c], b) : min(v[c], b);
                }
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < SKF_V; c++) { const int i = tid + 256 * c; if (i < NW) s_wmin[i] = v[c]; }
            __syncthreads();
        };
        if (w == 1) pass(s_h, 0, false);
        else {
            pass(s_h, 1, false);
            for (int d = 2; d < p2; d <<= 1) pass(s_wmin, d, false);
            if (w > p2) pass(s_wmin, w - p2, false);
        }
        if (t0 == 0 && tid == 0) {
            s_short = -1; s_abest = -1; s_am = NONE; s_ah = NONE;
            if (M <= T0) { // shorter than one window: only the last minimum (rightmost on ties)
                uint64_t m = NONE; int bp = -1;
                for (int p = max(0, M - w); p < M; p++) { const uint64_t h = s_h[p - e0]; if (h != NONE && h <= m) { m = h; bp = p; } }
                s_short = bp;
            } else {
                uint64_t m = NONE; int bp = -1;
                for (int p = T0 - w + 1; p <= T0 - 1; p++) { const uint64_t h = s_h[p - e0]; if (h != NONE && h <= m) { m = h; bp = p; } }
                s_am = m; s_abest = bp; s_ah = s_h[T0 - e0];
            }
        }
        __syncthreads();
        // an entry is reported iff it equals the minimum of one of the windows that contain it and end in [T0, M-1]; every such
        // minimum is <= the entry's hash, so the test is "sliding maximum of the (masked) window minima == hash", doubled the same way
        if (M > T0) {
            {
                // in place: every thread rewrites its own elements
#pragma unroll
                for (int c = 0; c < SKF_V; c++) {
                    const int i = tid + 256 * c, t = t0 + i;
                    if (i < NW && !(t >= T0 && t <= M - 1 && i < SKF_T + w - 1)) { s_wmin[i] = 0ull; v[c] = 0ull; }
                }
                __syncthreads();
            }
            if (w > 1) {
                for (int d = 1; d < p2; d <<= 1) pass(s_wmin, d, true);
                if (w > p2) pass(s_wmin, w - p2, true);
            }
        }
        for (int pi = tid; pi < SKF_T; pi += 256) {
            const int p = t0 + pi;
            if (p >= M) break;
            const uint64_t hp = s_h[pi + (w - 1)];
            if (hp == NONE) continue;
            bool e;
            if (M <= T0) e = (p == s_short);
            else {
                e = (p + w - 1 >= T0) && s_wmin[pi] == hp;
                if (p >= T0 - w + 1 && p <= T0 - 1 && s_am != NONE && hp == s_am) e = (p != s_abest) ? true : (s_ah > s_am);
            }
            if (e) emit(p);
        }
        __syncthreads();
    }
}
} // namespace
