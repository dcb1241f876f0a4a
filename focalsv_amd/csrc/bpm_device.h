// bpm_device.h -- device-side banded bit-parallel edit distance shared by K5 (verify), the rescue pass and K6 (path).
// Recurrence and end-site rule follow hifiasm-0.14 Levenshtein_distance.h:274-461.  bpm_run keeps the reference's 64-bit
// words (bands up to 63 rows: the doubled thresholds of the rescue pass); bpm_run32 is the same recurrence in 32-bit words
// for bands of at most 31 rows (k <= 15: every first-pass window), half the instructions on a 32-bit VALU.
#pragma once
#include "fsv_internal.h"

struct BpmState {
    uint64_t eq0, eq1, eq2, eq3; // match masks of the y rows inside the band
    uint64_t vp, vn;
};

__device__ __forceinline__ uint64_t bpm_pick_eq(const BpmState &s, uint32_t c)
{
    uint64_t lo = (c & 1u) ? s.eq1 : s.eq0;
    uint64_t hi = (c & 1u) ? s.eq3 : s.eq2;
    return (c & 2u) ? hi : lo;
}

__device__ __forceinline__ void bpm_eq_set(BpmState &s, uint32_t c, uint64_t bit)
{
    s.eq0 |= (c == 0u) ? bit : 0ull;
    s.eq1 |= (c == 1u) ? bit : 0ull;
    s.eq2 |= (c == 2u) ? bit : 0ull;
    s.eq3 |= (c == 3u) ? bit : 0ull;
}

// y base at padded-window column j (column 0 sits k bases before the predicted start);
// 4 = outside the read ('N' in the reference's fill_subregion).
__device__ __forceinline__ uint32_t bpm_ywin_base(const uint32_t *__restrict__ store, const fsv_wtask &t, int win0, int j)
{
    int p = win0 + j;
    if (p < 0 || p >= t.y_len) return 4u;
    return fsv_base_at(store, t.y_word, t.y_len, t.y_rev, p);
}

// determine_overlap_region (Correct.cpp:203-250): false = window geometrically impossible
__device__ __forceinline__ bool bpm_window_geometry(const fsv_wtask &t, fsv_wres &r, int k_cap = FSV_K_MAX)
{
    const int n = t.x_len, k = t.k, wlen = n + 2 * k;
    r.end_site = -1; r.err = -1; r.y_beg = -1; r.extra_begin = -1; r.extra_end = -1;
    if (t.y_start < 0 || t.y_len <= t.y_start || t.y_len - t.y_start + 2 * k + k_cap < wlen) return false;
    int ys = t.y_start - k, olen = min(wlen, t.y_len - ys);
    r.extra_end = (int16_t)(wlen - olen);
    r.extra_begin = 0;
    if (ys < 0) { r.extra_begin = (int16_t)(-ys); ys = 0; }
    r.y_beg = ys;
    return true;
}

// last-column scan, Levenshtein_distance.h:418-457
__device__ __forceinline__ int bpm_pick_end(const BpmState &s, int err, int n, int k, int &best_out)
{
    int best = -1, site = -1, e = err, ungapped = -1;
    if (e <= k) { best = e; site = n - 1; }
    for (int i = 0; i < 2 * k;) {
        e += (int)((s.vp >> i) & 1ull);
        e -= (int)((s.vn >> i) & 1ull);
        ++i;
        if (e <= k && (best < 0 || e <= best)) { best = e; site = n - 1 + i; }
        if (i == k) ungapped = e;
    }
    if (best >= 0 && k > 0 && ungapped == best) site = n - 1 + k;
    best_out = best;
    return best < 0 ? -1 : site;
}

// ---- 16-base stream fetch -----------------------------------------------------------------------------------
// 16 consecutive strand positions p..p+15 of a read as one word (base j in bits 2j..2j+1) plus a validity mask
// (bit j set when p+j lies inside the read).  All lanes of a wavefront fetch at the same loop trip, so the loads
// are uniform and can be issued one block ahead of their use; per-base global loads stalled every DP column.
struct Bases16 { uint32_t bits, valid; };

__device__ __forceinline__ uint32_t rev_fields2(uint32_t x)
{
    x = __brev(x);                                             // reverses bit order: field order reversed, bits inside a field swapped
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1); // swap the two bits of every field back
}

__device__ __forceinline__ uint32_t load_fwd16(const uint32_t *__restrict__ w, int nwords, int f)
{
    // forward bases f..f+15 (f may be negative or run past the end: missing words read as 0)
    const int a = f >> 4, sh = (f & 15) << 1;
    const uint32_t w0 = (a >= 0 && a < nwords) ? w[a] : 0u;
    const uint32_t w1 = (a + 1 >= 0 && a + 1 < nwords) ? w[a + 1] : 0u;
    return sh ? (w0 >> sh) | (w1 << (32 - sh)) : w0;
}

__device__ __forceinline__ uint32_t range_mask16(int p, int len)
{
    // bit j set iff 0 <= p + j < len
    const int lo = max(0, -p), hi = min(16, len - p);
    return hi > lo ? ((hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u)) : 0u;
}

__device__ __forceinline__ Bases16 fetch16(const uint32_t *__restrict__ store, uint32_t word_off, int len, int rev, int p)
{
    Bases16 r;
    const uint32_t *w = store + word_off;
    const int nwords = (len + 15) >> 4;
    r.valid = range_mask16(p, len);
    if (!rev) r.bits = load_fwd16(w, nwords, p);
    else r.bits = ~rev_fields2(load_fwd16(w, nwords, len - 16 - p)); // strand base j = complement of forward base len-1-p-j
    return r;
}

// x windows always lie inside their read: no validity, no strand
__device__ __forceinline__ uint32_t fetch16_x(const uint32_t *__restrict__ store, uint32_t word_off, int p)
{
    const uint32_t *w = store + word_off;
    const int a = p >> 4, sh = (p & 15) << 1;
    const uint32_t w0 = w[a], w1 = w[a + 1]; // the store keeps 4 words of slack behind the last read
    return sh ? (w0 >> sh) | (w1 << (32 - sh)) : w0;
}

// blockIdx -> work item.  The hardware deals consecutive workgroups to the eight XCDs in turn; taken as they come, the blocks that
// work on one read set -- consecutive items of a list -- land on all eight, and every XCD's 4 MB L2 has to hold the operands of all
// the sets in flight (k_chain fetched its lists five times over).  Here XCD x walks the x-th contiguous eighth of the n_blocks
// items.  The grid must cover 8 * ceil(n_blocks / 8) blocks; false: no item for this block.
__device__ __forceinline__ bool xcd_block(uint32_t n_blocks, uint32_t &item)
{
    const uint32_t per = (n_blocks + 7u) >> 3, b = blockIdx.x;
    if ((b >> 3) >= per) return false;
    item = (b & 7u) * per + (b >> 3);
    return item < n_blocks;
}

// ---- 64 bases at a time ---------------------------------------------------------------------------------------------------
// Five consecutive words of a read from word a (any dword alignment: global_load_dwordx4 + global_load_dword); words outside
// [0, nwords) read as 0.  One wide load where the 16-base fetches above issue eight narrow ones.
struct __attribute__((packed, aligned(4))) Words4 { uint32_t w[4]; };
__device__ __forceinline__ void load_words5(const uint32_t *__restrict__ w, int a, int nwords, uint32_t (&W)[5])
{
    if (a >= 0 && a + 4 < nwords) {
        const Words4 v = *reinterpret_cast<const Words4 *>(w + a);
        W[0] = v.w[0]; W[1] = v.w[1]; W[2] = v.w[2]; W[3] = v.w[3]; W[4] = w[a + 4];
    } else {
#pragma unroll
        for (int i = 0; i < 5; i++) { const int wi = a + i; W[i] = (wi >= 0 && wi < nwords) ? w[wi] : 0u; }
    }
}
// x windows lie inside their read and the store keeps 4 words of slack behind its last read: no bounds
__device__ __forceinline__ void load_words5_x(const uint32_t *__restrict__ w, int a, uint32_t (&W)[5])
{
    const Words4 v = *reinterpret_cast<const Words4 *>(w + a);
    W[0] = v.w[0]; W[1] = v.w[1]; W[2] = v.w[2]; W[3] = v.w[3]; W[4] = w[a + 4];
}
// the four 16-base blocks at strand positions p, p+16, p+32, p+48: what fetch16_x / fetch16 return for each of them
__device__ __forceinline__ void fetch64_x(const uint32_t *__restrict__ store, uint32_t word_off, int p, uint32_t (&xb)[4])
{
    uint32_t W[5];
    load_words5_x(store + word_off, p >> 4, W);
    const uint32_t sh = (uint32_t)(p & 15) << 1;
#pragma unroll
    for (int j = 0; j < 4; j++) xb[j] = __builtin_amdgcn_alignbit(W[j + 1], W[j], sh);
}
__device__ __forceinline__ void fetch64(const uint32_t *__restrict__ store, uint32_t word_off, int len, int rev, int p, uint32_t (&bits)[4], uint32_t (&valid)[4])
{
    uint32_t W[5];
    const int f = rev ? len - 64 - p : p;   // forward start of the lowest block (reverse strand: the mirror of the last one)
    load_words5(store + word_off, f >> 4, (len + 15) >> 4, W);
    const uint32_t sh = (uint32_t)(f & 15) << 1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        bits[j] = rev ? ~rev_fields2(__builtin_amdgcn_alignbit(W[4 - j], W[3 - j], sh)) : __builtin_amdgcn_alignbit(W[j + 1], W[j], sh);
        valid[j] = range_mask16(p + 16 * j, len);
    }
}

// Hooks for K6: `sink` sees every DP column (D0 and the post-update VP/VN).
struct BpmNoSink {
    __device__ __forceinline__ void operator()(int, uint64_t, uint64_t, uint64_t) const {}
};

// even bit positions of a 32-bit word -> 16-bit mask (bit j = bit 2j of the input)
__device__ __forceinline__ uint32_t compress_even16(uint32_t d)
{
    d &= 0x55555555u;
    d = (d | (d >> 1)) & 0x33333333u; d = (d | (d >> 2)) & 0x0f0f0f0fu; d = (d | (d >> 4)) & 0x00ff00ffu; d = (d | (d >> 8)) & 0xffffu;
    return d;
}

// One column of the recurrence (Levenshtein_distance.h:330-366); false when the column's diagonal cell is a mismatch.
__device__ __forceinline__ bool bpm_column(uint64_t eq, uint64_t &vp, uint64_t &vn, uint64_t &d0_out)
{
    const uint64_t x = eq | vn;
    const uint64_t d0 = ((vp + (x & vp)) ^ vp) | x;
    const uint64_t hn = vp & d0;
    const uint64_t hp = vn | ~(vp | d0);
    const uint64_t sh = d0 >> 1;
    vn = sh & hp;
    vp = hn | ~(sh | hp);
    d0_out = d0;
    return (d0 & 1ull) != 0ull;
}

template <class Sink>
__device__ __forceinline__ void bpm_run(const uint32_t *__restrict__ store, const fsv_wtask &t, fsv_wres &r, Sink sink)
{
    if (!bpm_window_geometry(t, r)) return;
    const int n = t.x_len, k = t.k;
    const int win0 = t.y_start - k;
    BpmState s;
    s.eq0 = s.eq1 = s.eq2 = s.eq3 = 0; s.vp = 0; s.vn = 0;
    for (int b16 = 0; b16 <= 2 * k; b16 += 16) {
        const Bases16 y = fetch16(store, t.y_word, t.y_len, t.y_rev, win0 + b16);
        for (int j = 0; j < 16 && b16 + j <= 2 * k; j++)
            if ((y.valid >> j) & 1u) bpm_eq_set(s, (y.bits >> (2 * j)) & 3u, 1ull << (b16 + j));
    }
    int err = 0;
    uint32_t xb = fetch16_x(store, t.x_word, t.x_start);
    Bases16 yb = fetch16(store, t.y_word, t.y_len, t.y_rev, win0 + 2 * k + 1);
    if (2 * k + 17 <= 64) {
        // The reference slides its four match masks one row per column (shift, then set the new top row).  Here the masks
        // of a 16-column block cover all 2k+17 y rows the block touches (bit b = row win0 + blk + b), so a column's masks are
        // a shift by j and a band mask, and the slide costs one shift per 16 columns: the same bits, a third fewer instructions.
        const uint64_t band = (2ull << (2 * k)) - 1ull;
        for (int blk = 0; blk < n; blk += 16) {
            // next block's operands are requested before this block's 16 columns are computed
            const uint32_t xn = fetch16_x(store, t.x_word, t.x_start + blk + 16);
            const Bases16 yn = fetch16(store, t.y_word, t.y_len, t.y_rev, win0 + 2 * k + 1 + blk + 16);
            {
                const uint32_t lo = compress_even16(yb.bits), hi = compress_even16(yb.bits >> 1), v = yb.valid & 0xffffu;
                s.eq0 |= (uint64_t)(~lo & ~hi & v) << (2 * k + 1);
                s.eq1 |= (uint64_t)(lo & ~hi & v) << (2 * k + 1);
                s.eq2 |= (uint64_t)(~lo & hi & v) << (2 * k + 1);
                s.eq3 |= (uint64_t)(lo & hi & v) << (2 * k + 1);
            }
            const int lim = min(16, n - blk);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                if (j < lim) {
                    const uint64_t eq = (bpm_pick_eq(s, (xb >> (2 * j)) & 3u) >> j) & band;
                    uint64_t d0;
                    if (!bpm_column(eq, s.vp, s.vn, d0)) {
                        ++err;
                        if (err - 2 * k > k) return; // Levenshtein_distance.h:367-375
                    }
                    sink(blk + j, d0, s.vp, s.vn);
                }
            }
            s.eq0 >>= 16; s.eq1 >>= 16; s.eq2 >>= 16; s.eq3 >>= 16;
            xb = xn; yb = yn;
        }
    } else {
        // bands wider than 47 rows (the doubled thresholds of the rescue pass): the reference's column-by-column slide
        const uint64_t top = 1ull << (2 * k);
        for (int blk = 0; blk < n; blk += 16) {
            const uint32_t xn = fetch16_x(store, t.x_word, t.x_start + blk + 16);
            const Bases16 yn = fetch16(store, t.y_word, t.y_len, t.y_rev, win0 + 2 * k + 1 + blk + 16);
            const int lim = min(16, n - blk);
            for (int j = 0; j < lim; j++) {
                const int i = blk + j;
                uint64_t d0;
                if (!bpm_column(bpm_pick_eq(s, (xb >> (2 * j)) & 3u), s.vp, s.vn, d0)) {
                    ++err;
                    if (err - 2 * k > k) return;
                }
                sink(i, d0, s.vp, s.vn);
                if (i + 1 < n) {
                    s.eq0 >>= 1; s.eq1 >>= 1; s.eq2 >>= 1; s.eq3 >>= 1;
                    if ((yb.valid >> j) & 1u) bpm_eq_set(s, (yb.bits >> (2 * j)) & 3u, top);
                }
            }
            xb = xn; yb = yn;
        }
    }
    int best;
    r.end_site = bpm_pick_end(s, err, n, k, best);
    r.err = best;
}

// ---- 32-bit form, k <= 15 -----------------------------------------------------------------------------------------------
// Why 32 bits give the reference's bits: the match masks are zero above the band (bit 2k), so above the band X == VN == 0
// throughout (VN = (D0 >> 1) & HP needs D0 to rise from 0 to 1 between neighbouring bits, and above the band D0 is the carry
// run of VP + (X & VP): ones from bit 2k+1 upwards, never a rising edge).  The only out-of-band bit that reaches the band is
// D0 bit 2k+1, which D0 >> 1 brings into the top row -- and it equals the carry out of bit 2k whatever VP holds up there
// ((VP + carry) ^ VP at that bit).  With 2k+1 <= 31 that bit exists in a 32-bit word; carries lost beyond bit 31 only
// feed bits beyond 31.  Checked bit for bit against the 64-bit path and the reference-minted vectors (tests/test_gpu_k5.py).
struct BpmNoSink32 {
    __device__ __forceinline__ void operator()(int, int, uint32_t, uint32_t, uint32_t, uint32_t) {}
    __device__ __forceinline__ void flush(int) {}
};

// Sink32 sees every column: (first column of the 16-column block, column inside the block -- a compile-time constant after
// unrolling --, D0, HP of the column, VP / VN after it).
template <class Sink>
__device__ __forceinline__ void bpm_run32(const uint32_t *__restrict__ store, const fsv_wtask &t, fsv_wres &r, Sink &sink)
{
    if (!bpm_window_geometry(t, r)) return;
    const int n = t.x_len, k = t.k;
    const int win0 = t.y_start - k;
    // y as two bit planes + validity over the 2k+17 rows a 16-column block touches (bit b = row win0 + blk + b): the match
    // mask of a column with x base c is XNOR(lo, c0) & XNOR(hi, c1) & valid, shifted down by the column's offset in the block
    uint64_t ylo = 0, yhi = 0, yv = 0;
    for (int b16 = 0; b16 <= 2 * k; b16 += 16) {
        const Bases16 y = fetch16(store, t.y_word, t.y_len, t.y_rev, win0 + b16);
        const uint32_t m = (1u << min(16, 2 * k + 1 - b16)) - 1u;
        ylo |= (uint64_t)(compress_even16(y.bits) & m) << b16;
        yhi |= (uint64_t)(compress_even16(y.bits >> 1) & m) << b16;
        yv |= (uint64_t)(y.valid & m) << b16;
    }
    uint32_t vp = 0, vn = 0;
    int err = 0;
    const uint32_t band = (2u << (2 * k)) - 1u;
    // Operands arrive 64 bases (five words) at a time, a chunk ahead of the columns that use them.  Fetched a 16-base block at a
    // time -- two words each of x and y every 16 columns -- a 64-byte sector of the store was touched sixteen times over ~3 000
    // instructions, and with 16 000 windows in flight per XCD a quarter of those touches missed the L2 again: the counters showed
    // 5.7 times the kernel's algorithmic bytes (profiles/r02_f_pmc_hbm_traffic.json before this change).
    const uint32_t *const xw = store + t.x_word, *const yw = store + t.y_word;
    const int ynw = (t.y_len + 15) >> 4, Y0 = win0 + 2 * k + 1;
    // forward start of chunk c: the strand position itself, or (reverse strand) the mirror of the chunk's last block
    const int yf0 = t.y_rev ? t.y_len - 64 - Y0 : Y0;
    const uint32_t xsh = (uint32_t)(t.x_start & 15) << 1, ysh = (uint32_t)(yf0 & 15) << 1;   // the same in every chunk
    uint32_t X[5], Y[5], XN[5] = {0, 0, 0, 0, 0}, YN[5] = {0, 0, 0, 0, 0};
    auto load_x = [&](int cb, uint32_t (&W)[5]) { load_words5_x(xw, (t.x_start + cb) >> 4, W); };
    auto load_y = [&](int cb, uint32_t (&W)[5]) { load_words5(yw, (t.y_rev ? yf0 - cb : yf0 + cb) >> 4, ynw, W); };
    load_x(0, X); load_y(0, Y);
    for (int cb = 0; cb < n; cb += 64) {
        if (cb + 64 < n) { load_x(cb + 64, XN); load_y(cb + 64, YN); }
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int blk = cb + 16 * b;
            if (blk < n) {
                const uint32_t xb = __builtin_amdgcn_alignbit(X[b + 1], X[b], xsh);
                // strand base j of the block = complement of forward base (block's forward start + 15 - j)
                const uint32_t ybits = t.y_rev ? ~rev_fields2(__builtin_amdgcn_alignbit(Y[4 - b], Y[3 - b], ysh)) : __builtin_amdgcn_alignbit(Y[b + 1], Y[b], ysh);
                const uint32_t yvalid = range_mask16(Y0 + blk, t.y_len);
                ylo |= (uint64_t)compress_even16(ybits) << (2 * k + 1);
                yhi |= (uint64_t)compress_even16(ybits >> 1) << (2 * k + 1);
                yv |= (uint64_t)(yvalid & 0xffffu) << (2 * k + 1);
                const int lim = min(16, n - blk);
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if (j < lim) {
                        const uint32_t c0 = (uint32_t)(((int32_t)(xb << (31 - 2 * j))) >> 31), c1 = (uint32_t)(((int32_t)(xb << (30 - 2 * j))) >> 31);
                        const uint32_t eq = ~((uint32_t)(ylo >> j) ^ c0) & ~((uint32_t)(yhi >> j) ^ c1) & (uint32_t)(yv >> j) & band;
                        const uint32_t x = eq | vn;
                        const uint32_t d0 = ((vp + (x & vp)) ^ vp) | x;
                        const uint32_t hn = vp & d0;
                        const uint32_t hp = vn | ~(vp | d0);
                        const uint32_t sh = d0 >> 1;
                        vn = sh & hp;
                        vp = hn | ~(sh | hp);
                        err += (int)(~d0 & 1u);
                        sink(blk, j, d0, hp, vp, vn);
                    }
                }
                // Levenshtein_distance.h:367-375 gives up at the first column where err - 2k > k; err never decreases, so looking once
                // per block ends in the same "no match" (and the end-site scan below could not find a row <= k either)
                if (err - 2 * k > k) return;
                ylo >>= 16; yhi >>= 16; yv >>= 16;
            }
        }
#pragma unroll
        for (int i = 0; i < 5; i++) { X[i] = XN[i]; Y[i] = YN[i]; }
    }
    sink.flush(n);
    BpmState s; s.eq0 = s.eq1 = s.eq2 = s.eq3 = 0; s.vp = vp; s.vn = vn;
    int best;
    r.end_site = bpm_pick_end(s, err, n, k, best);
    r.err = best;
}

// ---- wide bands: k <= 95 (191 rows) in six 32-bit limbs ----------------------------------------------------------------
// For reads whose windows differ by more than 8 % (BASELINE configs[4], ONT-profile reads: the reference hands those to
// Flye; its own BPM stops at 63 rows).  The same recurrence, end-site rule and early exit on a 192-bit word -- restated on
// 256 bits in oracle/bpm.c (orc_bpm_wide), which is checked against a plain banded DP and against the 64-bit functions where
// both apply.  Limb l holds band rows 32 l .. 32 l + 31.
#define FSV_WL 6
#define FSV_K_WIDE 95

struct WideNoSink {
    __device__ __forceinline__ void operator()(int, const uint32_t *, const uint32_t *, const uint32_t *) {}
};

// the end-site scan of bpm_pick_end on limbs
__device__ __forceinline__ int bpm_pick_end_wide(const uint32_t *vp, const uint32_t *vn, int err, int n, int k, int &best_out)
{
    int best = -1, site = -1, e = err, ungapped = -1;
    if (e <= k) { best = e; site = n - 1; }
    for (int i = 0; i < 2 * k;) {
        e += (int)((vp[i >> 5] >> (i & 31)) & 1u);
        e -= (int)((vn[i >> 5] >> (i & 31)) & 1u);
        ++i;
        if (e <= k && (best < 0 || e <= best)) { best = e; site = n - 1 + i; }
        if (i == k) ungapped = e;
    }
    if (best >= 0 && k > 0 && ungapped == best) site = n - 1 + k;
    best_out = best;
    return best < 0 ? -1 : site;
}

// Sink sees every column: (column, D0, HP of the column, VP after it), six limbs each.
template <class Sink>
__device__ __forceinline__ void bpm_run_wide(const uint32_t *__restrict__ store, const fsv_wtask &t, fsv_wres &r, Sink &sink, int k_cap)
{
    if (!bpm_window_geometry(t, r, k_cap)) return;
    const int n = t.x_len, k = t.k;
    const int win0 = t.y_start - k;
    // y planes over the 2k+17 rows of a 16-column block: seven limbs (bit b = row win0 + blk + b)
    uint32_t ylo[FSV_WL + 1], yhi[FSV_WL + 1], yv[FSV_WL + 1], vp[FSV_WL], vn[FSV_WL], bandm[FSV_WL];
#pragma unroll
    for (int l = 0; l <= FSV_WL; l++) { ylo[l] = yhi[l] = yv[l] = 0u; }
#pragma unroll
    for (int l = 0; l < FSV_WL; l++) {
        vp[l] = vn[l] = 0u;
        const int lo = 32 * l, nb = 2 * k + 1 - lo;     // band rows in this limb
        bandm[l] = nb >= 32 ? 0xffffffffu : (nb <= 0 ? 0u : ((1u << nb) - 1u));
    }
    // rows 0 .. 2k, 16 at a time: chunk c covers rows 16c .. 16c+15 = half of limb c >> 1
#pragma unroll
    for (int c = 0; c < 2 * FSV_WL; c++) {
        const int b16 = 16 * c;
        if (b16 <= 2 * k) {
            const Bases16 y = fetch16(store, t.y_word, t.y_len, t.y_rev, win0 + b16);
            const uint32_t m = (1u << min(16, 2 * k + 1 - b16)) - 1u;
            const int sh = (c & 1) * 16;
            ylo[c >> 1] |= (compress_even16(y.bits) & m) << sh;
            yhi[c >> 1] |= (compress_even16(y.bits >> 1) & m) << sh;
            yv[c >> 1] |= (y.valid & m) << sh;
        }
    }
    int err = 0;
    uint32_t xb = fetch16_x(store, t.x_word, t.x_start);
    Bases16 yb = fetch16(store, t.y_word, t.y_len, t.y_rev, win0 + 2 * k + 1);
    const int top = 2 * k + 1, tl = top >> 5, ts = top & 31;     // where a block's 16 new rows enter the planes
    for (int blk = 0; blk < n; blk += 16) {
        const uint32_t xn = fetch16_x(store, t.x_word, t.x_start + blk + 16);
        const Bases16 yn = fetch16(store, t.y_word, t.y_len, t.y_rev, win0 + 2 * k + 1 + blk + 16);
        {
            const uint64_t lo = (uint64_t)compress_even16(yb.bits) << ts, hi = (uint64_t)compress_even16(yb.bits >> 1) << ts,
                           v = (uint64_t)(yb.valid & 0xffffu) << ts;
#pragma unroll
            for (int l = 0; l <= FSV_WL; l++) {
                if (l == tl) { ylo[l] |= (uint32_t)lo; yhi[l] |= (uint32_t)hi; yv[l] |= (uint32_t)v; }
                if (l == tl + 1) { ylo[l] |= (uint32_t)(lo >> 32); yhi[l] |= (uint32_t)(hi >> 32); yv[l] |= (uint32_t)(v >> 32); }
            }
        }
        const int lim = min(16, n - blk);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (j < lim) {
                const uint32_t c0 = (uint32_t)(((int32_t)(xb << (31 - 2 * j))) >> 31), c1 = (uint32_t)(((int32_t)(xb << (30 - 2 * j))) >> 31);
                uint32_t d0[FSV_WL], hp[FSV_WL];
                uint32_t carry = 0u;
#pragma unroll
                for (int l = 0; l < FSV_WL; l++) {
                    // the column's match mask: the planes shifted down by j (bits from the limb above fill in)
                    const uint32_t pl = j ? (ylo[l] >> j | ylo[l + 1] << (32 - j)) : ylo[l], ph = j ? (yhi[l] >> j | yhi[l + 1] << (32 - j)) : yhi[l],
                                   pv = j ? (yv[l] >> j | yv[l + 1] << (32 - j)) : yv[l];
                    const uint32_t eq = ~(pl ^ c0) & ~(ph ^ c1) & pv & bandm[l];
                    const uint32_t x = eq | vn[l];
                    const uint32_t a = x & vp[l];
                    const uint32_t s1 = vp[l] + a, s2 = s1 + carry;
                    carry = (uint32_t)(s1 < a) | (uint32_t)(s2 < s1);
                    d0[l] = (s2 ^ vp[l]) | x;
                    hp[l] = vn[l] | ~(vp[l] | d0[l]);
                }
#pragma unroll
                for (int l = 0; l < FSV_WL; l++) {
                    const uint32_t hn = vp[l] & d0[l];
                    const uint32_t sh = d0[l] >> 1 | (l + 1 < FSV_WL ? d0[l + 1] << 31 : 0u);
                    vn[l] = sh & hp[l];
                    vp[l] = hn | ~(sh | hp[l]);
                }
                err += (int)(~d0[0] & 1u);
                sink(blk + j, d0, hp, vp);
            }
        }
        if (err - 2 * k > k) return;     // as bpm_run32: once per block, same outcome
#pragma unroll
        for (int l = 0; l <= FSV_WL; l++) {   // slide the planes 16 rows down
            const uint32_t up_lo = l < FSV_WL ? ylo[l + 1] : 0u, up_hi = l < FSV_WL ? yhi[l + 1] : 0u, up_v = l < FSV_WL ? yv[l + 1] : 0u;
            ylo[l] = ylo[l] >> 16 | up_lo << 16; yhi[l] = yhi[l] >> 16 | up_hi << 16; yv[l] = yv[l] >> 16 | up_v << 16;
        }
        xb = xn; yb = yn;
    }
    int best;
    r.end_site = bpm_pick_end_wide(vp, vn, err, n, k, best);
    r.err = best;
}
