// aln.hip -- contig-vs-reference-window alignment (fsv_align_batch) for gfx950.
//
// Replaces `minimap2 -a -x asm5 --cs -r2k` + pysam read-back (focalsv/4_sv_calling/Dippav/DipPAV_variant_call.py:103-112;
// extract_contig_signature_CCS.py:14-47, 342-432).  Restated in oracle/aln.c, which this file must match bit for bit.
//   k_sketch / k_uniq   seeds (ha_sketch without HPC, k = 19)           sketch.cpp:39-137
//   k_chain_aln         co-linear chains, one wavefront per pair, all state in LDS; the inverted piece as a record of the other strand
//   k_aln_events        gap-free runs vs DP events, padding, X-drop end extension
//   k_extract_boxes     an event of more than max_cells cells becomes a pair of its own, seeded / chained / walked again end to end
//   k_corner            what is still too large inside such a box: gap-free X-drop from its corners, the rest one I + one D
//   k_gap_shift         every gap of the stitched CIGAR to its leftmost position (minimap2's mm_fix_cigar rule)
//   k_nw                dual-affine global DP of one event on anti-diagonals (one workgroup per event), the
//                       recurrence / tie rules / backtrack of the in-tree ksw2 (ksw2_extz2_sse.c:171-196, ksw2.h:115-150)
#include "asm_kernels.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

#define ALN_AMAX 8192
#define ALN_CHAIN_STRIDE (ALN_AMAX + 8)   // anchors of one record slot (a box's chain carries its two fixed end pairs as well)
#define ALN_MAX_REC 5       // records per contig: up to ALN_MAJ_REC chains on its majority strand, the part behind a cut, one chain of the other strand
#define ALN_MAJ_REC 3       // the primary chain and up to two supplementary ones on the same strand
#define ALN_SUP_MIN 200     // chain score a supplementary chain needs
#define ALN_SUB_OCC 2       // an event larger than max_cells is seeded again: minimizers occurring at most this often in each side of its box
#define ALN_SUB_PER 1500    // ... with a minimizer window of max(w, L / 1500 + 1), L = the box's longer side
#define ALN_EV_CAP 2048      // events per pair
#define ALN_CG_CAP 1024      // CIGAR runs per event
#define NW_LDS_Q 3072        // query length up to which the rolling DP rows live in LDS (11 x 4 B x 3072 = 132 KB of the CU's 160)
#define NW_NEG (-(1 << 29))

struct AlnHeader { int32_t qbeg, tbeg, qend, tend, n_events, n_chain, rev, status; uint32_t ev_off, pad; };   // ev_off: first event in the packed list
struct AlnEvent { int32_t qs, qe, ts, te; };   // inclusive
struct NwTask { uint32_t pair; int32_t qs, ql, ts, tl; uint32_t cg_off; uint64_t bt_off; uint64_t row_off; uint32_t out_idx, pad; };

__device__ __forceinline__ int ilog2_u32(uint32_t v) { return 31 - __clz((int)v); }

// ------------------------------------------------------------------------------------------------ chain
// One block (one wavefront) per contig; its records go to the slots p0 .. p0 + R - 1 (pair_q / pair_t / hdr / chain are per slot).
//   SUB = false: a contig against its reference window (oracle/aln.c:orc_aln_chains).  Seeds unique in both sequences; the
//     best chain on the contig's majority strand is the primary record, what it leaves uncovered is chained again (up to
//     ALN_MAJ_REC chains); then the anchors of the other strand are chained once -- an inverted piece of the contig: minimap2
//     reports it as a reverse-strand supplementary record and breaks the alignment around it, and DipPAV's split rule only pairs
//     consecutive records of one strand (extract_contig_signature_CCS.py:286), so an inversion yields no call.  When that
//     chain's span on the contig lies strictly between the k-mers of two consecutive anchors of a majority chain (both parts
//     keeping min_anchors anchors) the majority chain is cut there and the part behind the cut becomes one more record.
//   SUB = true: the box of an event larger than max_cells, seeded again on its own (oracle/aln.c:sub_align): minimizers that
//     occur at most twice in each side (k_uniq with max_occ = 2), every pair of equal hash on the same strand an anchor, one
//     chain, written between the two fixed base pairs (-1, -1) and (lenq - 1, lent - 1) -- the outer chain's anchors.
template <bool SUB>
__global__ __launch_bounds__(64) void k_chain_aln(const int32_t *__restrict__ read_len,
                                                  const fsv_mz *__restrict__ mz, const uint32_t *__restrict__ mz_off,
                                                  const uint32_t *__restrict__ mz_cnt, const uint32_t *__restrict__ pair_q,
                                                  const uint32_t *__restrict__ pair_t, uint64_t *__restrict__ chain_out,
                                                  AlnHeader *__restrict__ hdr, uint32_t R, fsv_aln_params P)
{
    __shared__ uint64_t s_key[ALN_AMAX];
    __shared__ int32_t s_f[ALN_AMAX];
    __shared__ uint16_t s_aux[ALN_AMAX];
    __shared__ int s_cnt[ALN_MAX_REC];
    const int lane = threadIdx.x;
    const uint32_t p0 = blockIdx.x * R, rq = pair_q[p0], rt = pair_t[p0];
    const int lenq = read_len[rq], lent = read_len[rt];
    const int nq = (int)mz_cnt[rq], nt = (int)mz_cnt[rt];
    const fsv_mz *mq = mz + mz_off[rq] + nq, *mt = mz + mz_off[rt]; // contig: position-sorted copy, reference: hash-sorted
    AlnHeader h; h.qbeg = h.tbeg = h.qend = h.tend = 0; h.n_events = 0; h.n_chain = 0; h.rev = 0; h.status = 1; h.ev_off = 0; h.pad = 0;
    if (lane < (int)R) hdr[p0 + lane] = h;   // every slot starts out empty

    // chain DP over the n anchors in s_key, sorted by (contig, reference) coordinate: look back 64 anchors, the best-scoring
    // predecessor, the nearer one on ties (oracle/aln.c:chain_dp); returns the best chain's last anchor (the first on ties),
    // predecessors in s_aux (0xffff: none), scores in s_f.  A contig against its own reference window is co-linear except at
    // its SVs, so 64 anchors are settled at once as in the assembler's k_chain: hypothesis "every anchor links to its
    // predecessor" (the scores are a prefix sum), proof that no other predecessor beats it (cand(i, j) <= f[j] + k: only the j
    // with f[j] + k > f[i] are evaluated), one sequential step where it fails.  Round 1 walked the anchors one by one, a
    // barrier and three LDS round trips each (4.3 ms for 50 kb contigs).
    auto cand_of = [&](int qe, int te, int qj, int tj, int fj, bool &ok) -> int {
        const int dq = qe - qj, dt = te - tj;
        ok = false;
        if (dq <= 0 || dt <= 0) return 0;
        const int gap = dq > dt ? dq - dt : dt - dq;
        if (gap > P.max_gap) return 0;
        int sc = min(min(dq, dt), P.k);
        if (gap) sc -= (gap >> 7) + (ilog2_u32((uint32_t)gap) >> 1) + 1;
        ok = true;
        return sc + fj;
    };
    auto step_seq = [&](int i) {     // the DP step of anchor i as the sequential loop does it; all lanes take part
        const uint64_t ki = s_key[i];
        const int qe = (int)(ki >> 32), te = (int)(uint32_t)ki;
        const int j = i - 1 - lane;
        bool ok = false; int cand = 0;
        if (j >= 0) { const uint64_t kj = s_key[j]; cand = cand_of(qe, te, (int)(kj >> 32), (int)(uint32_t)kj, s_f[j], ok); }
        // scores can drop below zero here (gap penalty), so bias before packing; lanes without a legal predecessor sit out
        const int mine = ok ? ((cand + (1 << 20)) * 64 + (63 - lane)) : -1; // < 2^27
        const int bestp = wave_max_i32(mine);
        const int bests = bestp < 0 ? -(1 << 30) : (int)(bestp >> 6) - (1 << 20);
        if (bests > P.k) { if (mine == bestp) { s_f[i] = bests; s_aux[i] = (uint16_t)j; } }
        else if (lane == 0) { s_f[i] = P.k; s_aux[i] = 0xffff; }
        __syncthreads();
    };
    auto run_dp = [&](int n) -> int {
        if (n > 0) step_seq(0);
        for (int i0 = 1; i0 < n;) {
            const int nb = min(64, n - i0), i = i0 + lane;
            const bool in = lane < nb;
            int qe = 0, te = 0, sc1 = 0; bool ok1 = false;
            if (in) {
                const uint64_t ki = s_key[i], kp = s_key[i - 1];
                qe = (int)(ki >> 32); te = (int)(uint32_t)ki;
                sc1 = cand_of(qe, te, (int)(kp >> 32), (int)(uint32_t)kp, 0, ok1);
            }
            int pre = ok1 ? sc1 : 0;
            for (int off = 1; off < 64; off <<= 1) { const int o2 = __shfl_up(pre, off, 64); if (lane >= off) pre += o2; }
            const int fi = s_f[i0 - 1] + pre;
            bool bad = in && !(ok1 && fi > P.k);
            __syncthreads();
            if (in) s_f[i] = fi;
            __syncthreads();
            for (int d = 2; d <= 64; d++) {
                const int j = i - d;
                const bool live = in && !bad && j >= 0;
                if (!__any(live)) break;
                int fj = 0;
                if (live) fj = s_f[j];
                const bool need = live && fj + P.k > fi;
                if (__any(need)) {
                    if (need) {
                        bool ok2; const uint64_t kj = s_key[j];
                        const int c2 = cand_of(qe, te, (int)(kj >> 32), (int)(uint32_t)kj, fj, ok2);
                        if (ok2 && c2 > fi) bad = true;
                    }
                }
            }
            const uint64_t badm = __ballot(bad);
            const int good = badm ? (int)__ffsll((long long)badm) - 1 : nb;
            if (lane < good) s_aux[i] = (uint16_t)(i - 1);
            __syncthreads();
            i0 += good;
            if (good < nb) { step_seq(i0); i0++; }
        }
        long long bk = -1;
        for (int i = lane; i < n; i += 64) { long long v = (long long)s_f[i] * 16384 + (16383 - i); bk = v > bk ? v : bk; }
        bk = wave_max_i64(bk);
        return 16383 - (int)(bk & 16383);
    };
    // anchors of the chain that ends at `best`: how many, and the first one
    auto chain_len = [&](int best, int &cnt, int &first) {
        cnt = 0; first = best;
        if (lane == 0) { int c = best; while (c != 0xffff) { cnt++; first = c; c = s_aux[c]; } }
        cnt = __shfl(cnt, 0, 64); first = __shfl(first, 0, 64);
    };

    if (SUB) {
        int n = 0;
        for (int base = 0; base < nq; base += 64) {
            const int i = base + lane;
            int c = 0; uint64_t k0 = 0, k1 = 0;
            if (i < nq) {
                const fsv_mz a = mq[i];
                int l2 = 0, h2 = nt;
                while (l2 < h2) { int mid = (l2 + h2) >> 1; if (mt[mid].hash < a.hash) l2 = mid + 1; else h2 = mid; }
                for (int e = 0; e < 2 && l2 + e < nt; e++) {     // the list holds a hash at most twice, the lower position first
                    const fsv_mz b = mt[l2 + e];
                    if (b.hash != a.hash) break;
                    if (b.rev == a.rev) { const uint64_t key = (uint64_t)a.pos << 32 | b.pos; if (c == 0) k0 = key; else k1 = key; c++; }
                }
            }
            const uint64_t m1 = __ballot(c >= 1), m2 = __ballot(c >= 2), below = (1ull << lane) - 1;
            const int at = n + __popcll(m1 & below) + __popcll(m2 & below);
            if (c >= 1 && at < ALN_AMAX) s_key[at] = k0;
            if (c >= 2 && at + 1 < ALN_AMAX) s_key[at + 1] = k1;
            n += __popcll(m1) + __popcll(m2);
        }
        if (n > ALN_AMAX) n = ALN_AMAX;
        __syncthreads();
        int cnt = 0, first = 0, best = 0;
        if (n >= P.min_anchors) { best = run_dp(n); chain_len(best, cnt, first); }
        if (lane == 0) {
            uint64_t *out = chain_out + (size_t)p0 * ALN_CHAIN_STRIDE;
            int keep = 0;
            if (cnt >= P.min_anchors) {
                int c = best, k2 = cnt;
                while (c != 0xffff) { out[k2--] = s_key[c]; c = s_aux[c]; }
                keep = cnt;     // strictly in front of the box's last base pair
                while (keep > 0 && ((int)(out[keep] >> 32) >= lenq - 1 || (int)(uint32_t)out[keep] >= lent - 1)) keep--;
            }
            out[0] = ~0ull;                                                             // (-1, -1)
            out[keep + 1] = (uint64_t)(uint32_t)(lenq - 1) << 32 | (uint32_t)(lent - 1);
            h.n_chain = keep + 2; h.status = 0;
            hdr[p0] = h;
        }
        return;
    }

    int n = 0, nrev = 0, nfwd = 0;
    for (int base = 0; base < nq; base += 64) {
        int i = base + lane;
        bool hit = false; uint64_t key = 0; uint16_t aux = 0;
        if (i < nq) {
            fsv_mz a = mq[i];
            int l2 = 0, h2 = nt;
            while (l2 < h2) { int mid = (l2 + h2) >> 1; if (mt[mid].hash < a.hash) l2 = mid + 1; else h2 = mid; }
            if (l2 < nt) { fsv_mz b = mt[l2]; if (b.hash == a.hash) { hit = true; key = (uint64_t)a.pos << 32 | b.pos; aux = (uint16_t)(a.span | ((a.rev ^ b.rev) << 8)); } }
        }
        uint64_t m = __ballot(hit);
        int at = n + __popcll(m & ((1ull << lane) - 1));
        if (hit && at < ALN_AMAX) { s_key[at] = key; s_aux[at] = aux; }
        nrev += __popcll(__ballot(hit && (aux >> 8)));
        nfwd += __popcll(__ballot(hit && !(aux >> 8)));
        n += __popcll(m);
    }
    if (n > ALN_AMAX) n = ALN_AMAX;
    __syncthreads();
    const int rev = nrev > nfwd;
    h.rev = rev;
    // the majority strand's anchors stay in LDS (compacted in place); the others wait in the last slot's chain buffer
    uint64_t *tmp = chain_out + (size_t)(p0 + R - 1) * ALN_CHAIN_STRIDE;
    int m2 = 0, nb = 0;
    for (int base = 0; base < n; base += 64) {
        int i = base + lane;
        bool keep = false, oth = false; uint64_t key = 0;
        if (i < n) {
            const uint16_t aux = s_aux[i];
            const int strand = aux >> 8;
            key = s_key[i];
            // on the reverse strand the contig coordinate is that of the reverse complement
            if (strand) { int qp = (int)(key >> 32), span = aux & 0xff; qp = (lenq - 1) - (qp - span + 1); key = (uint64_t)(uint32_t)qp << 32 | (uint32_t)key; }
            keep = strand == rev; oth = !keep;
        }
        const uint64_t mk = __ballot(keep), mo = __ballot(oth), below = (1ull << lane) - 1;
        const int at = m2 + __popcll(mk & below), ao = nb + __popcll(mo & below);
        __syncthreads();
        if (keep) s_key[at] = key;
        if (oth) tmp[ao] = key;
        m2 += __popcll(mk); nb += __popcll(mo);
        __syncthreads();
    }
    n = m2;
    if (n < P.min_anchors) return;
    // anchors arrive in contig order; on the reverse strand the contig coordinate was mirrored, so the order is reversed
    if (rev) {
        for (int i = lane; i < n / 2; i += 64) { const uint64_t a = s_key[i], b = s_key[n - 1 - i]; s_key[i] = b; s_key[n - 1 - i] = a; }
        __syncthreads();
    }
    // The best chain is the primary alignment.  What it leaves uncovered is chained again (minimap2 reports such pieces as
    // supplementary alignments, and DipPAV calls the SVs beyond the chaining gap from consecutive records of one contig,
    // extract_contig_signature_CCS.py:251-327): the anchors inside the query interval of a chain are taken out -- they are a
    // contiguous run, the list is in query order -- and the rest goes through the same DP, up to ALN_MAJ_REC chains.
    int n_rec = 0;
    for (int rec = 0; rec < ALN_MAJ_REC && rec < (int)R && n >= P.min_anchors; rec++) {
        const int best = run_dp(n);
        int cnt, first;
        chain_len(best, cnt, first);
        if (cnt < P.min_anchors || (rec > 0 && s_f[best] < ALN_SUP_MIN)) break;
        if (lane == 0) {
            uint64_t *out = chain_out + (size_t)(p0 + rec) * ALN_CHAIN_STRIDE;
            int c = best, k2 = cnt;
            while (c != 0xffff) { out[--k2] = s_key[c]; c = s_aux[c]; }
            h.n_chain = cnt; h.status = 0;
            hdr[p0 + rec] = h;
            s_cnt[rec] = cnt;
        }
        n_rec++;
        // drop the anchors whose query coordinate lies in [q(first), q(best)]: indices lo .. hi of the sorted list
        const int qlo = (int)(s_key[first] >> 32), qhi = (int)(s_key[best] >> 32);
        int lo = n, hi = -1;
        for (int i = lane; i < n; i += 64) { const int qe = (int)(s_key[i] >> 32); if (qe >= qlo && qe <= qhi) { lo = min(lo, i); hi = max(hi, i); } }
        lo = -wave_max_i32(-lo); hi = wave_max_i32(hi);
        __syncthreads();
        const int tail = n - 1 - hi;
        for (int base = 0; base < tail; base += 64) {   // move the tail down, front to back (ascending: no overlap hazard inside a pass)
            const int i = base + lane;
            uint64_t v = 0;
            if (i < tail) v = s_key[hi + 1 + i];
            __syncthreads();
            if (i < tail) s_key[lo + i] = v;
            __syncthreads();
        }
        n = lo + tail;
    }
    // the other strand, once
    if (n_rec == 0 || n_rec >= (int)R || nb < P.min_anchors) return;
    const int mrev = !rev;
    __threadfence_block();      // tmp and the chains above were written with plain stores by other lanes
    __syncthreads();
    for (int i = lane; i < nb; i += 64) s_key[i] = tmp[mrev ? nb - 1 - i : i];
    __syncthreads();
    {
        const int best = run_dp(nb);
        int cnt, first;
        chain_len(best, cnt, first);
        if (cnt < P.min_anchors || s_f[best] < ALN_SUP_MIN) return;
        // span of the chain's k-mers on the contig, in the majority strand's coordinates
        const int lo_a = (lenq - 1) - (int)(s_key[best] >> 32), hi_a = (lenq - 1) - ((int)(s_key[first] >> 32) - P.k + 1);
        int cut_r = -1, cut_s = -1;
        for (int r = 0; r < n_rec && cut_r < 0; r++) {
            const uint64_t *c = chain_out + (size_t)(p0 + r) * ALN_CHAIN_STRIDE;
            const int cr = s_cnt[r];
            for (int base = P.min_anchors - 1; base + 1 + P.min_anchors <= cr; base += 64) {
                const int s = base + lane;
                const bool ok = s + 1 + P.min_anchors <= cr && (int)(c[s] >> 32) < lo_a && hi_a < (int)(c[s + 1] >> 32) - P.k + 1;
                const uint64_t m = __ballot(ok);
                if (m) { cut_r = r; cut_s = base + (int)__ffsll((long long)m) - 1; break; }
            }
        }
        if (cut_r >= 0 && n_rec + 1 < (int)R) {
            // the part behind the cut becomes the next record
            const int tail = s_cnt[cut_r] - (cut_s + 1);
            const uint64_t *src = chain_out + (size_t)(p0 + cut_r) * ALN_CHAIN_STRIDE + cut_s + 1;
            uint64_t *dst = chain_out + (size_t)(p0 + n_rec) * ALN_CHAIN_STRIDE;
            for (int i = lane; i < tail; i += 64) dst[i] = src[i];
            if (lane == 0) {
                h.rev = rev; h.status = 0;
                h.n_chain = cut_s + 1; hdr[p0 + cut_r] = h;
                h.n_chain = tail; hdr[p0 + n_rec] = h;
            }
            n_rec++;
        }
        if (lane == 0) {
            uint64_t *out = chain_out + (size_t)(p0 + n_rec) * ALN_CHAIN_STRIDE;
            int c = best, k2 = cnt;
            while (c != 0xffff) { out[--k2] = s_key[c]; c = s_aux[c]; }
            h.rev = mrev; h.n_chain = cnt; h.status = 0;
            hdr[p0 + n_rec] = h;
        }
    }
}

// ------------------------------------------------------------------------------------------------ seed thinning
// Windows beyond ~760 kb would need a minimizer window above 255 to keep a seed list below ALN_AMAX; instead w stays 255 and
// every m-th minimizer by hash survives, on the reference window and its contigs alike (oracle/aln.c does the same).  One
// block per sequence compacts its raw list in place (the order of a raw list is arbitrary: k_uniq sorts).
__global__ __launch_bounds__(256) void k_thin_seeds(fsv_mz *__restrict__ mz, const uint32_t *__restrict__ mz_off, uint32_t *__restrict__ mz_cnt,
                                                    const uint16_t *__restrict__ thin)
{
    __shared__ uint32_t s_n, s_base;
    const uint32_t r = blockIdx.x, m = thin[r];
    if (m <= 1) return;
    fsv_mz *a = mz + mz_off[r];
    const uint32_t n = min(mz_cnt[r], mz_off[r + 1] - mz_off[r]);
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 256) {
        const uint32_t i = base + threadIdx.x;
        fsv_mz v; v.hash = 0; v.pos = 0; v.rev = 0; v.span = 0; v.pad = 0;
        bool keep = false;
        if (i < n) { v = a[i]; keep = (v.hash >> 11) % (uint64_t)m == 0; }
        // rank inside the chunk: wave ballots, the waves' counts added up through LDS
        const uint64_t bal = __ballot(keep);
        const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        __shared__ uint32_t s_w[4];
        if (lane == 0) s_w[wv] = (uint32_t)__popcll(bal);
        __syncthreads();          // also: every thread has read its a[i] before anyone overwrites a slot at or below it
        uint32_t off = 0;
        for (uint32_t k2 = 0; k2 < wv; k2++) off += s_w[k2];
        if (threadIdx.x == 0) s_base = s_n;
        __syncthreads();
        if (keep) a[s_base + off + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = v;     // destination index <= i: only slots already read
        __syncthreads();
        if (threadIdx.x == 0) s_n = s_base + s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) mz_cnt[r] = s_n;
}

// ------------------------------------------------------------------------------------------------ events
// query base of the strand-oriented contig
__device__ __forceinline__ uint32_t qbase(const uint32_t *__restrict__ store, uint32_t qw, int lenq, int rev, int p) { return fsv_base_at(store, qw, lenq, rev, p); }
// target base of a reference window: 0-3, or 4 where the window has an N (nm: one word per 16 bases, bit j = base j is an N; nullptr:
// the batch has no N at all).  An N pairs with nothing (no padding, no gap-free run across it, no exact match) and costs
// FSV_SC_AMBI in a score, as in minimap2 (sc_ambi = 1: the alignment runs through a short run of N as 'M').  In the 2-bit store an
// N holds a base hashed from its position (k_pack_ascii): the seeds of an N run then look like random sequence, unique and
// matching nothing, where minimap2 skips k-mers with an N.
#define FSV_SC_AMBI 1
// the mask array starts four words into its buffer; the word in front of it says whether any window of the batch has an N at all
// (k_pack_ascii sets it): a kernel drops the mask at its start when there is none, and then pays nothing for it
#define FSV_NM_LEAD 4
__device__ __forceinline__ const uint32_t *nm_active(const uint32_t *__restrict__ nm) { return (nm && nm[-FSV_NM_LEAD]) ? nm : nullptr; }
__device__ __forceinline__ uint32_t tbase(const uint32_t *__restrict__ store, const uint32_t *__restrict__ nm, uint32_t tw, int p)
{
    if (nm && ((nm[tw + ((uint32_t)p >> 4)] >> ((uint32_t)p & 15u)) & 1u)) return 4u;
    return fsv_base_fwd(store, tw, p);
}
__device__ __forceinline__ int pair_score(uint32_t qb, uint32_t tb, const fsv_aln_params &P) { return tb > 3u ? -FSV_SC_AMBI : (qb == tb ? P.a : -P.b); }
__host__ __device__ __forceinline__ uint32_t n_substitute(uint32_t pos)
{
    uint32_t x = pos * 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13;
    return x >> 30;
}

// GLOBAL: the box of an oversize event (k_chain_aln<true>): the chain carries its two fixed end pairs, the walk runs from the
// box's first base pair to its last (no X-drop extension, no clips).
// An event of more than max_cells cells is not handed to the DP: it is listed with qs = -1 - qs, WITHOUT its padding (the box
// between the two anchors) -- the host has it seeded again (first level) or closed from its corners (k_corner, inside a box).
template <bool GLOBAL>
__global__ __launch_bounds__(256) void k_aln_events(const uint32_t *__restrict__ store, const uint32_t *__restrict__ nm_all, const uint32_t *__restrict__ word_off,
                                                    const int32_t *__restrict__ read_len, const uint32_t *__restrict__ pair_q,
                                                    const uint32_t *__restrict__ pair_t, const uint64_t *__restrict__ chain,
                                                    AlnHeader *__restrict__ hdr, AlnEvent *__restrict__ events, AlnEvent *__restrict__ packed,
                                                    uint32_t *__restrict__ n_packed, fsv_aln_params P)
{
    const uint32_t *nm = nm_active(nm_all);
    __shared__ uint8_t s_cls[ALN_CHAIN_STRIDE];
    const uint32_t p = blockIdx.x;
    AlnHeader h = hdr[p];
    if (h.status != 0) return;
    const uint32_t qw = word_off[pair_q[p]], tw = word_off[pair_t[p]];
    const int lenq = read_len[pair_q[p]], lent = read_len[pair_t[p]], rev = h.rev, nch = h.n_chain, nseg = nch - 1;
    const uint64_t *c = chain + (size_t)p * ALN_CHAIN_STRIDE;
    // segment classes between consecutive anchors: 0 identical, 1 few mismatches ('M'), 2 needs DP
    for (int s = threadIdx.x; s < nseg; s += blockDim.x) {
        const int q0 = (int)(c[s] >> 32), t0 = (int)(uint32_t)c[s], dq = (int)(c[s + 1] >> 32) - q0, dt = (int)(uint32_t)c[s + 1] - t0;
        uint8_t cls = 2;
        if (dq == dt) {
            int mm = 0;
            for (int k2 = 1; k2 <= dq && mm <= P.max_mm_run; k2++) mm += qbase(store, qw, lenq, rev, q0 + k2) != tbase(store, nm, tw, t0 + k2);
            cls = mm == 0 ? 0 : (mm <= P.max_mm_run ? 1 : 2);
        }
        s_cls[s] = cls;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
#define CQ(i) ((int)(c[i] >> 32))
#define CT(i) ((int)(uint32_t)c[i])
    int qbeg = 0, tbeg = 0, qend = lenq - 1, tend = lent - 1;
    if (!GLOBAL) {
        const int qs0 = CQ(0) - P.k + 1, ts0 = CT(0) - P.k + 1;
        int x = 0, best = 0, bi = 0;
        for (int i = 1; qs0 - i >= 0 && ts0 - i >= 0; i++) {
            x += pair_score(qbase(store, qw, lenq, rev, qs0 - i), tbase(store, nm, tw, ts0 - i), P);
            if (x > best) { best = x; bi = i; }
            if (best - x > P.xdrop) break;
        }
        qbeg = qs0 - bi; tbeg = ts0 - bi;
        x = 0; best = 0; bi = 0;
        for (int i = 1; CQ(nch - 1) + i < lenq && CT(nch - 1) + i < lent; i++) {
            x += pair_score(qbase(store, qw, lenq, rev, CQ(nch - 1) + i), tbase(store, nm, tw, CT(nch - 1) + i), P);
            if (x > best) { best = x; bi = i; }
            if (best - x > P.xdrop) break;
        }
        qend = CQ(nch - 1) + bi; tend = CT(nch - 1) + bi;
    }
    AlnEvent *ev = events + (size_t)p * ALN_EV_CAP;
    int ne = 0, mstart_q = qbeg, s = 0, status = 0;
    while (s < nseg) {
        if (s_cls[s] < 2) { s++; continue; }
        int e = s;
        while (e + 1 < nseg && s_cls[e + 1] == 2) e++;
        int eqs = CQ(s) + 1, eqe = CQ(e + 1), ets = CT(s) + 1, ete = CT(e + 1);
        int lp = 0, rp = 0;
        int lim_l = min(eqs - mstart_q, P.pad);
        while (lp < lim_l && qbase(store, qw, lenq, rev, eqs - 1 - lp) == tbase(store, nm, tw, ets - 1 - lp)) lp++;
        int lim_r = P.pad;
        if (eqe + lim_r > qend) lim_r = qend - eqe;
        if (ete + lim_r > tend) lim_r = tend - ete;
        { int nx = e + 1; while (nx < nseg && s_cls[nx] < 2) nx++; if (nx < nseg && eqe + lim_r > CQ(nx)) lim_r = CQ(nx) - eqe; }
        while (rp < lim_r && qbase(store, qw, lenq, rev, eqe + 1 + rp) == tbase(store, nm, tw, ete + 1 + rp)) rp++;
        if (ne >= ALN_EV_CAP) { status = FSV_ECAP; break; }
        if ((long long)(eqe - eqs + 1 + lp + rp) * (ete - ets + 1 + lp + rp) > P.max_cells) {
            ev[ne].qs = -1 - eqs; ev[ne].qe = eqe; ev[ne].ts = ets; ev[ne].te = ete; ne++;
        } else {
            eqs -= lp; ets -= lp; eqe += rp; ete += rp;
            ev[ne].qs = eqs; ev[ne].qe = eqe; ev[ne].ts = ets; ev[ne].te = ete; ne++;
        }
        mstart_q = eqe + 1;
        s = e + 1;
    }
#undef CQ
#undef CT
    h.qbeg = qbeg; h.tbeg = tbeg; h.qend = qend; h.tend = tend; h.n_events = ne; h.status = status;
    // the pair's events also go to a list packed over all pairs (any order; the header says where): one D2H copy for the batch
    if (status == 0 && ne > 0) {
        h.ev_off = atomicAdd(n_packed, (uint32_t)ne);
        for (int e = 0; e < ne; e++) packed[h.ev_off + e] = ev[e];
    }
    hdr[p] = h;
}

// ------------------------------------------------------------------------------------------------ oversize events
// The two sides of an event's box as sequences of their own (strand-oriented, word-aligned) in a second store, where the
// seeding / chaining / event kernels see them as an ordinary (contig, window) pair.  One thread per output word.
struct BoxSrc { uint32_t src_word; int32_t src_len, rev, start; };     // source read (word offset, length, strand), first base of the box side
__global__ __launch_bounds__(256) void k_extract_boxes(const uint32_t *__restrict__ src_store, const uint32_t *__restrict__ src_nm, const BoxSrc *__restrict__ box,
                                                       const uint32_t *__restrict__ word_off, const int32_t *__restrict__ read_len,
                                                       uint32_t n_seq, uint32_t total_words, uint32_t *__restrict__ words, uint32_t *__restrict__ nm_out)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= total_words) return;
    uint32_t lo = 0, hi = n_seq;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (word_off[mid] <= w) lo = mid; else hi = mid; }
    const BoxSrc b = box[lo];
    const int len = read_len[lo], b0 = (int)(w - word_off[lo]) * 16;
    uint32_t v = 0;
    for (int j = 0; j < 16 && b0 + j < len; j++) v |= fsv_base_at(src_store, b.src_word, b.src_len, b.rev, b.start + b0 + j) << (2 * j);
    words[w] = v;
    if (w == 0) nm_out[-FSV_NM_LEAD] = src_nm[-FSV_NM_LEAD];
    if (src_nm[-FSV_NM_LEAD]) {      // the N positions of the side (only a reference window has any; it is never read on the other strand)
        uint32_t m = 0;
        if (!b.rev) for (int j = 0; j < 16 && b0 + j < len; j++) { const int sp = b.start + b0 + j; m |= ((src_nm[b.src_word + ((uint32_t)sp >> 4)] >> ((uint32_t)sp & 15u)) & 1u) << j; }
        nm_out[w] = m;
    }
}

// An event of a box's inner walk that is still larger than max_cells (nothing in it could be seeded: unrelated sequence, or a
// tandem array of short units) is closed from its two corners: gap-free X-drop extensions, the rest one insertion + one
// deletion (oracle/aln.c:corner_event).  One wavefront per event, 64 positions per step: the running score is a prefix sum,
// the running best a prefix maximum, the first position where best - score > xdrop ends the run.
struct CornerTask { uint32_t pair; int32_t qs, ql, ts, tl; };
__global__ __launch_bounds__(64) void k_corner(const uint32_t *__restrict__ store, const uint32_t *__restrict__ nm_all, const uint32_t *__restrict__ word_off,
                                               const int32_t *__restrict__ read_len, const uint32_t *__restrict__ pair_q,
                                               const uint32_t *__restrict__ pair_t, const AlnHeader *__restrict__ hdr,
                                               const CornerTask *__restrict__ tasks, int2 *__restrict__ out, fsv_aln_params P)
{
    const uint32_t *nm = nm_active(nm_all);
    const CornerTask T = tasks[blockIdx.x];
    const int lane = threadIdx.x;
    const uint32_t qw = word_off[pair_q[T.pair]], tw = word_off[pair_t[T.pair]];
    const int lenq = read_len[pair_q[T.pair]], rev = hdr[T.pair].rev;
    const int lim = min(T.ql, T.tl);
    int res[2] = {0, 0};
    for (int side = 0; side < 2; side++) {
        const int n = side == 0 ? lim : lim - res[0];
        int x0 = 0, best = 0, len = 0;       // score and best score in front of the current step, bases taken at the best
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            int d = 0;
            if (i < n) {
                const int qp = side == 0 ? T.qs + i : T.qs + T.ql - 1 - i, tp = side == 0 ? T.ts + i : T.ts + T.tl - 1 - i;
                d = pair_score(qbase(store, qw, lenq, rev, qp), tbase(store, nm, tw, tp), P);
            }
            int x = d;
            for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(x, off, 64); if (lane >= off) x += o; }
            x += x0;
            int m = i < n ? x : -(1 << 30);
            for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(m, off, 64); if (lane >= off) m = max(m, o); }
            m = max(m, best);
            const uint64_t brk = __ballot(i < n && m - x > P.xdrop);
            const int last = brk ? (int)__ffsll((long long)brk) - 1 : 63;      // lanes up to here are part of the run
            const int mb = __shfl(m, last, 64);                                // the best up to the end of the run (or of the step)
            if (mb > best) {
                const uint64_t at = __ballot(i < n && lane <= last && x == mb);
                len = base + (int)__ffsll((long long)at);                       // first position that reaches it
                best = mb;
            }
            if (brk) break;
            x0 = __shfl(x, 63, 64);
        }
        res[side] = len;
    }
    if (lane == 0) out[blockIdx.x] = make_int2(res[0], res[1]);
}

// ------------------------------------------------------------------------------------------------ gap left alignment
// minimap2 moves every I/D of a finished CIGAR to its leftmost position (mm_fix_cigar: a deletion while the reference base
// entering the gap on the left equals the one leaving it on the right, an insertion the same on the query) -- restated in
// oracle/aln.c:shift_gaps_left.  How far a gap could travel depends on the sequence alone, so the device answers that for
// all gaps of the batch at once (one wavefront per gap, 64 positions per step); the host then applies the shifts in CIGAR
// order, each bounded by the M run in front of it.
struct GapQuery { uint32_t slot; int32_t is_ins, off, len, cap; };
__global__ __launch_bounds__(64) void k_gap_shift(const uint32_t *__restrict__ store, const uint32_t *__restrict__ nm_all, const uint32_t *__restrict__ word_off,
                                                  const int32_t *__restrict__ read_len, const uint32_t *__restrict__ pair_q,
                                                  const uint32_t *__restrict__ pair_t, const AlnHeader *__restrict__ hdr,
                                                  const GapQuery *__restrict__ gaps, int32_t *__restrict__ max_shift)
{
    const uint32_t *nm = nm_active(nm_all);
    const GapQuery g = gaps[blockIdx.x];
    const int lane = threadIdx.x;
    const uint32_t qw = word_off[pair_q[g.slot]], tw = word_off[pair_t[g.slot]];
    const int lenq = read_len[pair_q[g.slot]], rev = hdr[g.slot].rev;
    int res = g.cap;
    for (int base = 0; base < g.cap; base += 64) {
        const int l = base + lane;
        bool stop = true;
        if (l < g.cap) {
            const int a = g.off - 1 - l, b = g.off + g.len - 1 - l;
            stop = g.is_ins ? qbase(store, qw, lenq, rev, a) != qbase(store, qw, lenq, rev, b)
                            : tbase(store, nm, tw, a) != tbase(store, nm, tw, b);
        }
        const uint64_t m = __ballot(stop);
        if (m) { res = base + (int)__ffsll((long long)m) - 1; break; }
    }
    if (lane == 0) max_shift[blockIdx.x] = res;
}

// ------------------------------------------------------------------------------------------------ NW
// One workgroup per event, cells of one anti-diagonal in parallel.  Rolling rows indexed by the query position:
//   H on diagonals d-1 and d-2, and the E/F/E2/F2 values *leaving* each cell of diagonal d-1.
// Per cell one traceback byte with ksw2's layout (ksw2.h:115-118), stored diagonal-major (byte (i+j)*ql + j: the cells of
// a diagonal are adjacent, so a wave's stores coalesce); k_nw_rows, whose target side is the short one, stores byte
// (i+j)*tl + i instead -- a 30 kb insertion against 50 reference bases then takes 1.5 MB of traceback, not 900.
struct NwRows {
    int32_t *rows; int stride;
    __device__ __forceinline__ int32_t *H(int d) const { return rows + (size_t)(((d) % 3 + 3) % 3) * stride; }
    __device__ __forceinline__ int32_t *E(int d) const { return rows + (size_t)(3 + (d & 1)) * stride; }
    __device__ __forceinline__ int32_t *F(int d) const { return rows + (size_t)(5 + (d & 1)) * stride; }
    __device__ __forceinline__ int32_t *E2(int d) const { return rows + (size_t)(7 + (d & 1)) * stride; }
    __device__ __forceinline__ int32_t *F2(int d) const { return rows + (size_t)(9 + (d & 1)) * stride; }
};

// cell (i, j) of diagonal d = i + j (ksw_extz2's recurrence, ksw2_extz2_sse.c; boundary: a gap of length l before the
// first cell costs min(q + e*l, q2 + e2*l))
// BYROW: the rolling rows are indexed by the target row i instead of the query column j (k_nw_rows: events with a short target side)
template <bool BYROW = false>
__device__ __forceinline__ void nw_cell(const NwRows &R, int d, int i, int j, uint32_t tb, uint32_t qb, bool two, const fsv_aln_params &P,
                                        uint8_t *__restrict__ bt, int bt_stride)
{
    const int xc = BYROW ? i : j;             // this cell's slot
    const int xd = xc - 1;                    // the diagonal neighbour (i-1, j-1)
    const int xe = BYROW ? i - 1 : j;         // E comes from (i-1, j)
    const int xf = BYROW ? i : j - 1;         // F comes from (i, j-1)
    int32_t hdiag, a, b, a2 = NW_NEG, b2 = NW_NEG;
    if (i == 0 && j == 0) hdiag = 0;
    else if (i == 0) { int g1 = -(P.q + P.e * j), g2 = two ? -(P.q2 + P.e2 * j) : NW_NEG; hdiag = max(g1, g2); }
    else if (j == 0) { int g1 = -(P.q + P.e * i), g2 = two ? -(P.q2 + P.e2 * i) : NW_NEG; hdiag = max(g1, g2); }
    else hdiag = R.H(d - 2)[xd];
    if (i == 0) {
        int g1 = -(P.q + P.e * (j + 1)), g2 = two ? -(P.q2 + P.e2 * (j + 1)) : NW_NEG;
        const int hup = max(g1, g2); // H(-1, j)
        a = hup - P.q - P.e; if (two) a2 = hup - P.q2 - P.e2;
    } else { a = R.E(d - 1)[xe]; if (two) a2 = R.E2(d - 1)[xe]; }
    if (j == 0) {
        int g1 = -(P.q + P.e * (i + 1)), g2 = two ? -(P.q2 + P.e2 * (i + 1)) : NW_NEG;
        const int hleft = max(g1, g2); // H(i, -1)
        b = hleft - P.q - P.e; if (two) b2 = hleft - P.q2 - P.e2;
    } else { b = R.F(d - 1)[xf]; if (two) b2 = R.F2(d - 1)[xf]; }
    int32_t h = hdiag + pair_score(qb, tb, P);
    uint8_t dd = 0;
    if (a > h) { h = a; dd = 1; }
    if (b > h) { h = b; dd = 2; }
    if (two && a2 > h) { h = a2; dd = 3; }
    if (two && b2 > h) { h = b2; dd = 4; }
    int32_t o = h - P.q;
    if (a > o) { dd |= 0x08; R.E(d)[xc] = a - P.e; } else R.E(d)[xc] = o - P.e;
    if (b > o) { dd |= 0x10; R.F(d)[xc] = b - P.e; } else R.F(d)[xc] = o - P.e;
    if (two) {
        o = h - P.q2;
        if (a2 > o) { dd |= 0x20; R.E2(d)[xc] = a2 - P.e2; } else R.E2(d)[xc] = o - P.e2;
        if (b2 > o) { dd |= 0x40; R.F2(d)[xc] = b2 - P.e2; } else R.F2(d)[xc] = o - P.e2;
    }
    R.H(d)[xc] = h;
    bt[(size_t)d * bt_stride + xc] = dd;    // diagonal-major; within a diagonal by column, or by row for a short target side
}

// ksw_backtrack (ksw2.h:120-150) by thread 0, emitted end-to-start then reversed in place.  The traceback bytes come through
// an LDS tile of NW_TD diagonals x NW_TC columns that the whole workgroup loads around the current cell: one memory latency
// per >= 16 steps instead of one per step.
#define NW_TD 32
#define NW_TC 32
template <bool BYROW = false>
__device__ __forceinline__ void nw_backtrack(const uint8_t *__restrict__ bt, int ql, int tl, uint32_t *__restrict__ cg, uint32_t *__restrict__ cg_n_out,
                                             uint8_t (*s_bt)[NW_TC], int *s_walk)
{
    const int bt_stride = BYROW ? tl : ql;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) { s_walk[0] = tl - 1; s_walk[1] = ql - 1; s_walk[2] = 0; s_walk[3] = 0; s_walk[4] = 0; }
    __syncthreads();
    for (;;) {
        const int i0 = s_walk[0], j0 = s_walk[1];
        if (i0 < 0 || j0 < 0) break;
        const int dtop = i0 + j0, jtop = j0;    // tile: diagonals dtop-NW_TD+1 .. dtop, columns jtop-NW_TC+1 .. jtop
        for (int idx = tid; idx < NW_TD * NW_TC; idx += nt) {
            const int rd = idx / NW_TC, rc = idx % NW_TC;
            const int d = dtop - rd, j = jtop - rc, i = d - j;
            s_bt[rd][rc] = (d >= 0 && j >= 0 && i >= 0 && i < tl) ? bt[(size_t)d * bt_stride + (BYROW ? i : j)] : (uint8_t)0;
        }
        __syncthreads();
        if (tid == 0) {
            int i = i0, j = j0, state = s_walk[2], n = s_walk[3];
            bool over = s_walk[4] != 0;
            while (i >= 0 && j >= 0) {
                const int rd = dtop - (i + j), rc = jtop - j;
                if (rd >= NW_TD || rc >= NW_TC) break;   // left the tile
                const uint8_t dd = s_bt[rd][rc];
                if (state == 0) state = dd & 7;
                else if (!((dd >> (state + 2)) & 1)) state = 0;
                if (state == 0) state = dd & 7;
                uint32_t op;
                if (state == 0) { op = 0; i--; j--; }
                else if (state == 1 || state == 3) { op = 2; i--; }
                else { op = 1; j--; }
                if (n && (cg[n - 1] & 0xf) == op) cg[n - 1] += 1u << 4;
                else if (n < ALN_CG_CAP) cg[n++] = 1u << 4 | op;
                else over = true;
            }
            s_walk[0] = i; s_walk[1] = j; s_walk[2] = state; s_walk[3] = n; s_walk[4] = over ? 1 : 0;
        }
        __syncthreads();
    }
    if (tid != 0) return;
    int i = s_walk[0], j = s_walk[1], n = s_walk[3];
    bool over = s_walk[4] != 0;
    auto put = [&](uint32_t op, uint32_t len) {
        if (n && (cg[n - 1] & 0xf) == op) cg[n - 1] += len << 4;
        else if (n < ALN_CG_CAP) cg[n++] = len << 4 | op;
        else over = true;
    };
    if (i >= 0) put(2, (uint32_t)(i + 1));
    if (j >= 0) put(1, (uint32_t)(j + 1));
    for (int k2 = 0; k2 < n / 2; k2++) { uint32_t t = cg[k2]; cg[k2] = cg[n - 1 - k2]; cg[n - 1 - k2] = t; }
    *cg_n_out = over ? 0xffffffffu : (uint32_t)n;
}

// Queries up to QCAP bases: rolling rows in LDS (11 x QCAP x 4 B), thread t owns the query columns t, t+NT, ... for the whole
// sweep, so its query bases sit in registers; the target bases of the rows the sweep is crossing sit in an LDS ring that the
// workgroup refills every CH diagonals.  No global load on the per-diagonal critical path.
template <int QCAP, int NT>
__global__ __launch_bounds__(NT) void k_nw(const uint32_t *__restrict__ store, const uint32_t *__restrict__ nm_all, const uint32_t *__restrict__ word_off,
                                           const int32_t *__restrict__ read_len, const uint32_t *__restrict__ pair_q,
                                           const uint32_t *__restrict__ pair_t, const AlnHeader *__restrict__ hdr,
                                           const NwTask *__restrict__ tasks, uint8_t *__restrict__ bt_all,
                                           uint32_t *__restrict__ cg_all, uint32_t *__restrict__ cg_n, int32_t *__restrict__ scores, fsv_aln_params P)
{
    const uint32_t *nm = nm_active(nm_all);
    constexpr int C = QCAP / NT;
    constexpr int CH = QCAP >= 1024 ? 256 : 64;
    constexpr int TB = QCAP >= 2048 ? 4096 : 2 * QCAP;   // ring of target bases (power of two >= QCAP + CH)
    __shared__ int32_t s_rows[11 * QCAP];
    __shared__ uint8_t s_t[TB];
    __shared__ uint8_t s_bt[NW_TD][NW_TC];
    __shared__ int s_walk[8];
    const NwTask T = tasks[blockIdx.x];
    const uint32_t qw = word_off[pair_q[T.pair]], tw = word_off[pair_t[T.pair]];
    const int lenq = read_len[pair_q[T.pair]], rev = hdr[T.pair].rev;
    const int ql = T.ql, tl = T.tl, tid = threadIdx.x;
    const bool two = P.q2 >= 0;
    const NwRows R{s_rows, QCAP};
    uint8_t *bt = bt_all + T.bt_off;
    uint32_t qb[C];
#pragma unroll
    for (int m = 0; m < C; m++) { const int j = tid + m * NT; qb[m] = j < ql ? qbase(store, qw, lenq, rev, T.qs + j) : 0u; }
    for (int d = 0; d <= ql + tl - 2; d++) {
        if (d % CH == 0) {
            // rows d .. d+CH-1 enter the sweep during the next CH diagonals; rows below d-ql+1 have left it
            for (int i = d + tid; i < min(d + CH, tl); i += NT) s_t[i & (TB - 1)] = (uint8_t)tbase(store, nm, tw, T.ts + i);
            __syncthreads();
        }
#pragma unroll
        for (int m = 0; m < C; m++) {
            const int j = tid + m * NT, i = d - j;
            if (j < ql && i >= 0 && i < tl) nw_cell(R, d, i, j, s_t[i & (TB - 1)], qb[m], two, P, bt, ql);
        }
        // the next diagonal needs this one's LDS rows, not its traceback bytes: __syncthreads() would also drain the global
        // stores (vmcnt), a memory round trip per diagonal; they are drained once, by the barrier in front of the backtrack
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) scores[T.out_idx] = R.H(ql + tl - 2)[ql - 1];
    nw_backtrack(bt, ql, tl, cg_all + T.cg_off, cg_n + T.out_idx, s_bt, s_walk);
}

// The transposed case -- a long query against a short target (an insertion: ~50 reference bases of padding against the inserted
// kilobases): thread t owns the target rows t, t+NT, ..., the rolling rows are indexed by row, the query bases come through the LDS
// ring.  With the rows of such an event in ONE wavefront a diagonal costs a single-wave barrier; round 1 sent these events to
// k_nw<3072, 1024>, where each of their ~2 000 diagonals paid a 16-wave barrier for a few dozen cells (7.3 ms per bench step).
template <int RCAP, int NT>
__global__ __launch_bounds__(NT) void k_nw_rows(const uint32_t *__restrict__ store, const uint32_t *__restrict__ nm_all, const uint32_t *__restrict__ word_off,
                                                const int32_t *__restrict__ read_len, const uint32_t *__restrict__ pair_q,
                                                const uint32_t *__restrict__ pair_t, const AlnHeader *__restrict__ hdr,
                                                const NwTask *__restrict__ tasks, uint8_t *__restrict__ bt_all,
                                                uint32_t *__restrict__ cg_all, uint32_t *__restrict__ cg_n, int32_t *__restrict__ scores, fsv_aln_params P)
{
    const uint32_t *nm = nm_active(nm_all);
    constexpr int C = RCAP / NT;
    constexpr int CH = 64;
    constexpr int QB = 2 * RCAP;        // ring of query bases (power of two >= RCAP + CH)
    __shared__ int32_t s_rows[11 * RCAP];
    __shared__ uint8_t s_q[QB];
    __shared__ uint8_t s_bt[NW_TD][NW_TC];
    __shared__ int s_walk[8];
    const NwTask T = tasks[blockIdx.x];
    const uint32_t qw = word_off[pair_q[T.pair]], tw = word_off[pair_t[T.pair]];
    const int lenq = read_len[pair_q[T.pair]], rev = hdr[T.pair].rev;
    const int ql = T.ql, tl = T.tl, tid = threadIdx.x;
    const bool two = P.q2 >= 0;
    const NwRows R{s_rows, RCAP};
    uint8_t *bt = bt_all + T.bt_off;
    uint32_t tb[C];
#pragma unroll
    for (int m = 0; m < C; m++) { const int i = tid + m * NT; tb[m] = i < tl ? tbase(store, nm, tw, T.ts + i) : 0u; }
    for (int d = 0; d <= ql + tl - 2; d++) {
        if (d % CH == 0) {
            // columns d .. d+CH-1 enter the sweep during the next CH diagonals; columns below d-tl+1 have left it
            for (int j = d + tid; j < min(d + CH, ql); j += NT) s_q[j & (QB - 1)] = (uint8_t)qbase(store, qw, lenq, rev, T.qs + j);
            __syncthreads();
        }
#pragma unroll
        for (int m = 0; m < C; m++) {
            const int i = tid + m * NT, j = d - i;
            if (i < tl && j >= 0 && j < ql) nw_cell<true>(R, d, i, j, tb[m], s_q[j & (QB - 1)], two, P, bt, tl);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) scores[T.out_idx] = R.H(ql + tl - 2)[tl - 1];
    nw_backtrack<true>(bt, ql, tl, cg_all + T.cg_off, cg_n + T.out_idx, s_bt, s_walk);
}

// Any query length: rolling rows in HBM, bases fetched per cell (events with queries above NW_LDS_Q bases: rare, slow path)
__global__ __launch_bounds__(256) void k_nw_any(const uint32_t *__restrict__ store, const uint32_t *__restrict__ nm_all, const uint32_t *__restrict__ word_off,
                                                const int32_t *__restrict__ read_len, const uint32_t *__restrict__ pair_q,
                                                const uint32_t *__restrict__ pair_t, const AlnHeader *__restrict__ hdr,
                                                const NwTask *__restrict__ tasks, uint8_t *__restrict__ bt_all, int32_t *__restrict__ rows_all,
                                                uint32_t *__restrict__ cg_all, uint32_t *__restrict__ cg_n, int32_t *__restrict__ scores, fsv_aln_params P)
{
    const uint32_t *nm = nm_active(nm_all);
    __shared__ uint8_t s_bt[NW_TD][NW_TC];
    __shared__ int s_walk[8];
    const NwTask T = tasks[blockIdx.x];
    const uint32_t qw = word_off[pair_q[T.pair]], tw = word_off[pair_t[T.pair]];
    const int lenq = read_len[pair_q[T.pair]], rev = hdr[T.pair].rev;
    const int ql = T.ql, tl = T.tl;
    const bool two = P.q2 >= 0;
    const NwRows R{rows_all + T.row_off, ql};
    uint8_t *bt = bt_all + T.bt_off;
    for (int d = 0; d <= ql + tl - 2; d++) {
        const int jlo = max(0, d - (tl - 1)), jhi = min(ql - 1, d);
        for (int j = jlo + (int)threadIdx.x; j <= jhi; j += blockDim.x) {
            const int i = d - j;
            nw_cell(R, d, i, j, tbase(store, nm, tw, T.ts + i), qbase(store, qw, lenq, rev, T.qs + j), two, P, bt, ql);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x == 0) scores[T.out_idx] = R.H(ql + tl - 2)[ql - 1];
    nw_backtrack(bt, ql, tl, cg_all + T.cg_off, cg_n + T.out_idx, s_bt, s_walk);
}

// ------------------------------------------------------------------------------------------------ host side
struct DevBuf { void *p = nullptr; size_t cap = 0; };

struct AlnWs {
    DevBuf store, nmask, ascii, asc_off, word_off, len, wper, pair_q, pair_t, sk_ends, sk_low, sk_high, mz, mz_off, mz_cnt, warn, chain, hdr, events, ev_packed, ev_count, tasks, bt, rows, cg, cg_n, scores, gaps, gap_shift, thin, box_src, corner, corner_out;
    fsv_aln_stats stats;
    // the size classes of the event DP run side by side: a class is a handful of long-running blocks, never a full chip
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[3] = {nullptr, nullptr, nullptr};
    AlnWs *sub = nullptr;       // the workspace of the boxes of oversize events (a second, smaller alignment pass)
    const uint32_t *nm() const { return (const uint32_t *)nmask.p + FSV_NM_LEAD; }     // the N mask of the store (the kernels drop it when the batch has no N: nm_active)
    std::vector<DevBuf *> all() { return {&store, &nmask, &ascii, &asc_off, &word_off, &len, &wper, &pair_q, &pair_t, &sk_ends, &sk_low, &sk_high, &mz, &mz_off, &mz_cnt, &warn, &chain, &hdr, &events, &ev_packed, &ev_count, &tasks, &bt, &rows, &cg, &cg_n, &scores, &gaps, &gap_shift, &thin, &box_src, &corner, &corner_out}; }
};

void aln_ws_release(AlnWs *w)
{
    if (!w) return;
    if (w->sub) aln_ws_release(w->sub);
    for (DevBuf *b : w->all()) if (b->p) (void)hipFree(b->p);
    for (int i = 0; i < 3; i++) { if (w->side[i]) (void)hipStreamDestroy(w->side[i]); if (w->join[i]) (void)hipEventDestroy(w->join[i]); }
    if (w->fork) (void)hipEventDestroy(w->fork);
    delete w;
}

void aln_ws_free(fsv_ctx *ctx)
{
    aln_ws_release((AlnWs *)ctx->aln_ws);
    ctx->aln_ws = nullptr;
}

AlnWs *aln_ws_get(fsv_ctx *ctx)
{
    if (!ctx->aln_ws) { ctx->aln_ws = new AlnWs(); ctx->aln_ws_free = aln_ws_free; memset(&((AlnWs *)ctx->aln_ws)->stats, 0, sizeof(fsv_aln_stats)); }
    return (AlnWs *)ctx->aln_ws;
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
    TRY(ensure(ctx, b, std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (!v.empty()) FSV_HIP(ctx, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return FSV_OK;
}
struct Timer {
    std::chrono::steady_clock::time_point t0; fsv_ctx *ctx;
    explicit Timer(fsv_ctx *c) : ctx(c) { (void)hipStreamSynchronize(c->stream); t0 = std::chrono::steady_clock::now(); }
    double stop() { (void)hipStreamSynchronize(ctx->stream); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// packs pairs (query, target) into a store; returns lens / offsets
// ASCII -> 2-bit store on the device: one thread per output word, 16 source bytes each.  An N (anything but ACGT) gets a base hashed
// from its position and its bit in the mask word; the word in front of the mask array says whether the batch has any (nm_active)
__global__ __launch_bounds__(256) void k_pack_ascii(const char *__restrict__ ascii, const uint64_t *__restrict__ asc_off,
                                                    const uint32_t *__restrict__ word_off, const int32_t *__restrict__ read_len,
                                                    uint32_t n_reads, uint32_t total_words, uint32_t *__restrict__ words, uint32_t *__restrict__ nm_out)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= total_words) return;
    uint32_t lo = 0, hi = n_reads;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (word_off[mid] <= w) lo = mid; else hi = mid; }
    const uint32_t r = lo;
    const int len = read_len[r];
    const int b0 = (int)(w - word_off[r]) * 16;
    const char *src = ascii + asc_off[r] + b0;
    uint32_t v = 0, m = 0;
    for (int j = 0; j < 16 && b0 + j < len; j++) {
        const char c = src[j];
        uint32_t code = (c == 'C' || c == 'c') ? 1u : (c == 'G' || c == 'g') ? 2u : (c == 'T' || c == 't') ? 3u : 0u;
        if (code == 0u && c != 'A' && c != 'a') { code = n_substitute((uint32_t)(b0 + j)); m |= 1u << j; }     // N (anything else): see tbase
        v |= code << (2 * j);
    }
    words[w] = v;
    nm_out[w] = m;
    if (m) atomicOr(&nm_out[-FSV_NM_LEAD], 1u);      // (rare: a window with an N)
}

// sequences (already concatenated by the caller in two buffers: reference windows, contigs) -> device 2-bit store
int pack_pairs(fsv_ctx *ctx, AlnWs &W, const std::vector<const char *> &seq, const std::vector<uint64_t> &slen, std::vector<uint32_t> &word_off,
               std::vector<int32_t> &len, const char *dev_src = nullptr, uint32_t dev_first = 0)
{
    const uint32_t n = (uint32_t)seq.size();
    word_off.assign(n + 1, 0); len.resize(n);
    std::vector<uint64_t> asc_off(n + 1, 0);
    uint64_t w = 0;
    for (uint32_t r = 0; r < n; r++) {
        word_off[r] = (uint32_t)w; len[r] = (int32_t)slen[r];
        w += (slen[r] + 15) / 16;
        asc_off[r + 1] = asc_off[r] + slen[r];
    }
    if (w + 8 >= (1ull << 32)) return fsv_fail(ctx, FSV_EUNSUP, "alignment batch too large; split it");
    word_off[n] = (uint32_t)w;
    TRY(ensure(ctx, W.ascii, asc_off[n] + 64));
    // consecutive sequences that are adjacent in host memory go up in one copy (the callers pass two contiguous buffers)
    // sequences dev_first.. are already on the device, back to back at dev_src (contigs of the last assembly): one D2D copy
    const uint32_t n_host = dev_src ? dev_first : n;
    if (dev_src && n > dev_first)
        FSV_HIP(ctx, hipMemcpyAsync((char *)W.ascii.p + asc_off[dev_first], dev_src, asc_off[n] - asc_off[dev_first], hipMemcpyDeviceToDevice, ctx->stream));
    for (uint32_t r = 0; r < n_host;) {
        uint32_t e = r + 1;
        while (e < n_host && seq[e] == seq[e - 1] + slen[e - 1]) e++;
        FSV_HIP(ctx, hipMemcpyAsync((char *)W.ascii.p + asc_off[r], seq[r], asc_off[e] - asc_off[r], hipMemcpyHostToDevice, ctx->stream));
        r = e;
    }
    TRY(upload(ctx, W.asc_off, asc_off));
    TRY(upload(ctx, W.word_off, word_off));
    TRY(upload(ctx, W.len, len));
    TRY(ensure(ctx, W.store, (w + 8) * 4));
    FSV_HIP(ctx, hipMemsetAsync((uint32_t *)W.store.p + w, 0, 32, ctx->stream));
    TRY(ensure(ctx, W.nmask, (w + 8 + FSV_NM_LEAD) * 4));
    FSV_HIP(ctx, hipMemsetAsync(W.nmask.p, 0, FSV_NM_LEAD * 4, ctx->stream));      // the "some window has an N" word (set on the device: the host never looks at the text)
    FSV_HIP(ctx, hipMemsetAsync((uint32_t *)W.nmask.p + FSV_NM_LEAD + w, 0, 32, ctx->stream));
    hipLaunchKernelGGL(k_pack_ascii, dim3(fsv_grid_for(w, 256)), dim3(256), 0, ctx->stream, (const char *)W.ascii.p, (const uint64_t *)W.asc_off.p,
                       (const uint32_t *)W.word_off.p, (const int32_t *)W.len.p, n, (uint32_t)w, (uint32_t *)W.store.p, (uint32_t *)W.nmask.p + FSV_NM_LEAD);
    FSV_HIP(ctx, hipGetLastError());
    return FSV_OK;
}

// the event DP's size classes: 0 a short query, 3 a short target side under a long query (insertions: the row-owning
// single-wave kernel), 1 queries whose rolling rows fit in LDS, 2 the rest
inline int nw_class(int ql, int tl) { return ql <= 256 ? 0 : tl <= 256 ? 3 : ql <= NW_LDS_Q ? 1 : 2; }
inline uint64_t nw_bt_bytes(int ql, int tl) { return (uint64_t)(ql + tl) * (uint64_t)(nw_class(ql, tl) == 3 ? tl : ql); }   // diagonal-major traceback bytes

int run_nw(fsv_ctx *ctx, AlnWs &W, const std::vector<NwTask> &tasks, uint64_t bt_bytes, uint64_t row_words, const fsv_aln_params &P)
{
    // size classes, each a contiguous range of the (stably) reordered task array; outputs keep their task index
    const size_t n = tasks.size();
    std::vector<uint32_t> order;
    order.reserve(n);
    size_t cls_end[4];
    for (int c = 0; c < 4; c++) {
        for (size_t i = 0; i < n; i++)
            if (nw_class(tasks[i].ql, tasks[i].tl) == c) order.push_back((uint32_t)i);
        cls_end[c] = order.size();
    }
    std::vector<NwTask> sorted(n);
    for (size_t i = 0; i < n; i++) { sorted[i] = tasks[order[i]]; sorted[i].cg_off = order[i] * (uint32_t)ALN_CG_CAP; sorted[i].out_idx = order[i]; }
    TRY(upload(ctx, W.tasks, sorted));
    TRY(ensure(ctx, W.bt, bt_bytes + 16));
    TRY(ensure(ctx, W.rows, row_words * 4 + 16));
    TRY(ensure(ctx, W.cg, n * (size_t)ALN_CG_CAP * 4));
    TRY(ensure(ctx, W.cg_n, n * 4));
    TRY(ensure(ctx, W.scores, n * 4));
    if (!W.fork) {
        FSV_HIP(ctx, hipEventCreateWithFlags(&W.fork, hipEventDisableTiming));
        for (int i = 0; i < 3; i++) {
            FSV_HIP(ctx, hipStreamCreateWithFlags(&W.side[i], hipStreamNonBlocking));
            FSV_HIP(ctx, hipEventCreateWithFlags(&W.join[i], hipEventDisableTiming));
        }
    }
    // class c runs on its own stream between a fork and a join on the context's stream (class 0 stays on it)
    FSV_HIP(ctx, hipEventRecord(W.fork, ctx->stream));
    auto lane = [&](int c) -> hipStream_t { return c == 0 ? ctx->stream : W.side[c - 1]; };
    bool used[4] = {false, false, false, false};
    for (int c = 1; c < 4; c++)
        if (cls_end[c] > cls_end[c - 1]) { used[c] = true; FSV_HIP(ctx, hipStreamWaitEvent(W.side[c - 1], W.fork, 0)); }
    if (cls_end[0])
        hipLaunchKernelGGL((k_nw<256, 64>), dim3((uint32_t)cls_end[0]), dim3(64), 0, lane(0), (const uint32_t *)W.store.p, W.nm(), (const uint32_t *)W.word_off.p,
                           (const int32_t *)W.len.p, (const uint32_t *)W.pair_q.p, (const uint32_t *)W.pair_t.p, (const AlnHeader *)W.hdr.p,
                           (const NwTask *)W.tasks.p, (uint8_t *)W.bt.p, (uint32_t *)W.cg.p, (uint32_t *)W.cg_n.p, (int32_t *)W.scores.p, P);
    FSV_HIP(ctx, hipGetLastError());
    if (used[1])
        hipLaunchKernelGGL((k_nw<NW_LDS_Q, 1024>), dim3((uint32_t)(cls_end[1] - cls_end[0])), dim3(1024), 0, lane(1), (const uint32_t *)W.store.p, W.nm(),
                           (const uint32_t *)W.word_off.p, (const int32_t *)W.len.p, (const uint32_t *)W.pair_q.p, (const uint32_t *)W.pair_t.p,
                           (const AlnHeader *)W.hdr.p, (const NwTask *)W.tasks.p + cls_end[0], (uint8_t *)W.bt.p, (uint32_t *)W.cg.p,
                           (uint32_t *)W.cg_n.p, (int32_t *)W.scores.p, P);
    FSV_HIP(ctx, hipGetLastError());
    if (used[2])
        hipLaunchKernelGGL(k_nw_any, dim3((uint32_t)(cls_end[2] - cls_end[1])), dim3(256), 0, lane(2), (const uint32_t *)W.store.p, W.nm(),
                           (const uint32_t *)W.word_off.p, (const int32_t *)W.len.p, (const uint32_t *)W.pair_q.p, (const uint32_t *)W.pair_t.p,
                           (const AlnHeader *)W.hdr.p, (const NwTask *)W.tasks.p + cls_end[1], (uint8_t *)W.bt.p, (int32_t *)W.rows.p, (uint32_t *)W.cg.p,
                           (uint32_t *)W.cg_n.p, (int32_t *)W.scores.p, P);
    FSV_HIP(ctx, hipGetLastError());
    if (used[3])
        hipLaunchKernelGGL((k_nw_rows<256, 64>), dim3((uint32_t)(cls_end[3] - cls_end[2])), dim3(64), 0, lane(3), (const uint32_t *)W.store.p, W.nm(),
                           (const uint32_t *)W.word_off.p, (const int32_t *)W.len.p, (const uint32_t *)W.pair_q.p, (const uint32_t *)W.pair_t.p,
                           (const AlnHeader *)W.hdr.p, (const NwTask *)W.tasks.p + cls_end[2], (uint8_t *)W.bt.p, (uint32_t *)W.cg.p,
                           (uint32_t *)W.cg_n.p, (int32_t *)W.scores.p, P);
    FSV_HIP(ctx, hipGetLastError());
    for (int c = 1; c < 4; c++)
        if (used[c]) { FSV_HIP(ctx, hipEventRecord(W.join[c - 1], W.side[c - 1])); FSV_HIP(ctx, hipStreamWaitEvent(ctx->stream, W.join[c - 1], 0)); }
    return FSV_OK;
}

void push_cg(std::vector<uint32_t> &cg, uint32_t op, uint32_t len)
{
    if (!len) return;
    if (!cg.empty() && (cg.back() & 0xf) == op) cg.back() += len << 4;
    else cg.push_back(len << 4 | op);
}

// What one pass aligns: n_refs targets then n_pairs queries, already packed in W.store (word_off / len uploaded), pair p =
// (query n_refs + p, target pair_t[p]).
struct PassIn {
    uint32_t n_refs = 0, n_pairs = 0;
    std::vector<uint32_t> word_off; std::vector<int32_t> len;
    std::vector<uint8_t> wper; std::vector<uint16_t> thin;   // minimizer window / thinning factor per sequence
    std::vector<uint32_t> pair_t;
    std::vector<int32_t> pre_status;                           // per pair: != 0 -> not aligned, reported as is
};
// per record slot (n_pairs x R): header, status, and the ops between the record's first and last aligned base (no clips)
struct PassOut { uint32_t R = 1; std::vector<AlnHeader> hdr; std::vector<int32_t> status; std::vector<std::vector<uint32_t>> body; };

// seed window of a pair whose longer side has L bases (oracle/aln.c:seed_window): w = max(w0, L / per + 1); beyond 255 (the
// sketch kernels' limit) w stays 255 and every thin-th minimizer by hash survives, on both sides alike
inline void seed_window(int w0, uint64_t L, uint64_t per, uint8_t &w_out, uint16_t &thin_out)
{
    uint64_t w = std::max<uint64_t>((uint64_t)w0, L / per + 1);
    thin_out = 1;
    if (w > 255) { thin_out = (uint16_t)std::min<uint64_t>((w + 254) / 255, 65535); w = 255; }
    w_out = (uint8_t)w;
}

// seeds -> chains -> events -> DP -> the ops of every record.  depth 0: contigs against their windows (ALN_MAX_REC record
// slots per contig, X-drop ends); depth 1: the boxes of the first pass's oversize events, end to end.
int align_pass(fsv_ctx *ctx, AlnWs &W, const PassIn &S, const fsv_aln_params &P, int depth, PassOut &O, fsv_aln_stats *stats)
{
    const uint32_t np = S.n_pairs, nr = S.n_refs + S.n_pairs, n_refs = S.n_refs;
    const uint32_t R = depth == 0 ? ALN_MAX_REC : 1, ns = np * R;
    const std::vector<uint32_t> &word_off = S.word_off; const std::vector<int32_t> &len = S.len;
    O.R = R;
    Timer tseed(ctx);
    auto tr0 = std::chrono::steady_clock::now();
    auto trace = [&](const char *what) { if (getenv("FSV_TRACE")) { (void)hipStreamSynchronize(ctx->stream); auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[fsv] align%s %-14s %.2f ms\n", depth ? " (boxes)" : "", what, std::chrono::duration<double, std::milli>(t - tr0).count()); tr0 = t; } };
    TRY(upload(ctx, W.wper, S.wper));
    // kernels index pairs by record slot
    {
        std::vector<uint32_t> sq(ns), st(ns);
        for (uint32_t p = 0; p < np; p++) for (uint32_t r = 0; r < R; r++) { sq[p * R + r] = n_refs + p; st[p * R + r] = S.pair_t[p]; }
        TRY(upload(ctx, W.pair_q, sq));
        TRY(upload(ctx, W.pair_t, st));
    }
    std::vector<uint32_t> mz_off(nr + 1, 0);
    uint64_t m = 0;
    for (uint32_t r = 0; r < nr; r++) { mz_off[r] = (uint32_t)m; m += (uint64_t)len[r] + 64; } // worst case one minimizer per base
    if (m >= (1ull << 32)) return fsv_fail(ctx, FSV_EUNSUP, "alignment batch too large; split it");
    mz_off[nr] = (uint32_t)m;
    TRY(upload(ctx, W.mz_off, mz_off));
    TRY(ensure(ctx, W.mz, m * sizeof(fsv_mz)));
    TRY(ensure(ctx, W.mz_cnt, (size_t)nr * 4));
    TRY(ensure(ctx, W.warn, (size_t)nr * 4));
    FSV_HIP(ctx, hipMemsetAsync(W.warn.p, 0, (size_t)nr * 4, ctx->stream));
    FSV_HIP(ctx, hipMemsetAsync(W.mz_cnt.p, 0, (size_t)nr * 4, ctx->stream));
    if (P.k & 1) {
        const size_t total_words = word_off[nr];
        TRY(ensure(ctx, W.sk_ends, (total_words * 16 + 64) * 4));
        TRY(ensure(ctx, W.sk_low, (total_words + nr + 8) * 4));
        TRY(ensure(ctx, W.sk_high, (total_words + nr + 8) * 4));
        FSV_HIP(ctx, hipMemsetAsync(W.sk_low.p, 0, (total_words + nr + 8) * 4, ctx->stream));
        FSV_HIP(ctx, hipMemsetAsync(W.sk_high.p, 0, (total_words + nr + 8) * 4, ctx->stream));
        hipLaunchKernelGGL(k_sketch_fast, dim3(nr), dim3(256), 0, ctx->stream, (const uint32_t *)W.store.p, (const uint32_t *)W.word_off.p,
                           (const int32_t *)W.len.p, (const uint32_t *)W.mz_off.p, (fsv_mz *)W.mz.p, (uint32_t *)W.mz_cnt.p, nr, P.w, P.k, 0,
                           (uint32_t *)W.warn.p, (const uint8_t *)W.wper.p, (uint32_t *)W.sk_ends.p, (uint32_t *)W.sk_low.p, (uint32_t *)W.sk_high.p,
                           (const uint32_t *)nullptr);
    } else {
        uint32_t max_words = 1; int w_max = 1;
        for (uint32_t r = 0; r < nr; r++) { max_words = std::max<uint32_t>(max_words, (uint32_t)((len[r] + 15) / 16)); w_max = std::max<int>(w_max, S.wper[r]); }
        const uint32_t lds_words = std::min<uint32_t>(max_words, 8192u);
        FSV_HIP(ctx, hipFuncSetAttribute((const void *)k_sketch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sketch_lds_bytes(w_max, lds_words)));
        hipLaunchKernelGGL(k_sketch, dim3(nr), dim3(64), sketch_lds_bytes(w_max, lds_words), ctx->stream, (const uint32_t *)W.store.p,
                           (const uint32_t *)W.word_off.p, (const int32_t *)W.len.p, (const uint32_t *)W.mz_off.p, (fsv_mz *)W.mz.p,
                           (uint32_t *)W.mz_cnt.p, nr, P.w, P.k, 0, (uint32_t *)W.warn.p, (const uint8_t *)W.wper.p, w_max, lds_words);
    }
    FSV_HIP(ctx, hipGetLastError());
    bool any_thin = false;
    for (uint32_t r = 0; r < nr; r++) any_thin |= S.thin[r] > 1;
    if (any_thin) {
        TRY(upload(ctx, W.thin, S.thin));
        hipLaunchKernelGGL(k_thin_seeds, dim3(nr), dim3(256), 0, ctx->stream, (fsv_mz *)W.mz.p, (const uint32_t *)W.mz_off.p, (uint32_t *)W.mz_cnt.p,
                           (const uint16_t *)W.thin.p);
        FSV_HIP(ctx, hipGetLastError());
    }
    // first pass: seeds unique in their sequence; boxes: seeds that occur at most twice in their side
    hipLaunchKernelGGL(k_uniq<ALN_AMAX>, dim3(nr), dim3(256), 0, ctx->stream, (fsv_mz *)W.mz.p, (const uint32_t *)W.mz_off.p, (uint32_t *)W.mz_cnt.p,
                       (uint32_t *)W.warn.p, (const uint32_t *)nullptr, 0u, 0xffffffffu, (unsigned long long *)nullptr, depth == 0 ? 1u : (uint32_t)ALN_SUB_OCC);
    FSV_HIP(ctx, hipGetLastError());
    trace("sketch+uniq");
    if (stats) stats->ms_seed = tseed.stop();
    Timer tchain(ctx);
    TRY(ensure(ctx, W.chain, (size_t)ns * ALN_CHAIN_STRIDE * 8));
    TRY(ensure(ctx, W.hdr, (size_t)ns * sizeof(AlnHeader)));
    TRY(ensure(ctx, W.events, (size_t)ns * ALN_EV_CAP * sizeof(AlnEvent)));
    if (depth == 0)
        hipLaunchKernelGGL(k_chain_aln<false>, dim3(np), dim3(64), 0, ctx->stream, (const int32_t *)W.len.p,
                           (const fsv_mz *)W.mz.p, (const uint32_t *)W.mz_off.p, (const uint32_t *)W.mz_cnt.p, (const uint32_t *)W.pair_q.p,
                           (const uint32_t *)W.pair_t.p, (uint64_t *)W.chain.p, (AlnHeader *)W.hdr.p, R, P);
    else
        hipLaunchKernelGGL(k_chain_aln<true>, dim3(np), dim3(64), 0, ctx->stream, (const int32_t *)W.len.p,
                           (const fsv_mz *)W.mz.p, (const uint32_t *)W.mz_off.p, (const uint32_t *)W.mz_cnt.p, (const uint32_t *)W.pair_q.p,
                           (const uint32_t *)W.pair_t.p, (uint64_t *)W.chain.p, (AlnHeader *)W.hdr.p, R, P);
    FSV_HIP(ctx, hipGetLastError());
    trace("chain");
    if (stats) stats->ms_chain = tchain.stop();
    Timer tev(ctx);
    TRY(ensure(ctx, W.ev_packed, (size_t)ns * ALN_EV_CAP * sizeof(AlnEvent)));
    TRY(ensure(ctx, W.ev_count, 16));
    FSV_HIP(ctx, hipMemsetAsync(W.ev_count.p, 0, 4, ctx->stream));
    if (depth == 0)
        hipLaunchKernelGGL(k_aln_events<false>, dim3(ns), dim3(256), 0, ctx->stream, (const uint32_t *)W.store.p, W.nm(), (const uint32_t *)W.word_off.p,
                           (const int32_t *)W.len.p, (const uint32_t *)W.pair_q.p, (const uint32_t *)W.pair_t.p, (const uint64_t *)W.chain.p,
                           (AlnHeader *)W.hdr.p, (AlnEvent *)W.events.p, (AlnEvent *)W.ev_packed.p, (uint32_t *)W.ev_count.p, P);
    else
        hipLaunchKernelGGL(k_aln_events<true>, dim3(ns), dim3(256), 0, ctx->stream, (const uint32_t *)W.store.p, W.nm(), (const uint32_t *)W.word_off.p,
                           (const int32_t *)W.len.p, (const uint32_t *)W.pair_q.p, (const uint32_t *)W.pair_t.p, (const uint64_t *)W.chain.p,
                           (AlnHeader *)W.hdr.p, (AlnEvent *)W.events.p, (AlnEvent *)W.ev_packed.p, (uint32_t *)W.ev_count.p, P);
    FSV_HIP(ctx, hipGetLastError());
    std::vector<AlnHeader> &hdr = O.hdr;
    hdr.assign(ns, AlnHeader());
    uint32_t n_ev = 0;
    FSV_HIP(ctx, hipMemcpyAsync(hdr.data(), W.hdr.p, (size_t)ns * sizeof(AlnHeader), hipMemcpyDeviceToHost, ctx->stream));
    FSV_HIP(ctx, hipMemcpyAsync(&n_ev, W.ev_count.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<AlnEvent> events(n_ev);
    if (n_ev) {
        FSV_HIP(ctx, hipMemcpyAsync(events.data(), W.ev_packed.p, (size_t)n_ev * sizeof(AlnEvent), hipMemcpyDeviceToHost, ctx->stream));
        FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    trace("events");
    if (stats) stats->ms_events = tev.stop();
    // DP tasks; an oversize event (qs < 0) is a box for the second pass (depth 0) or a corner task (inside a box)
    Timer tdp(ctx);
    std::vector<NwTask> tasks;
    struct Big { uint32_t slot; int32_t qs, qe, ts, te; };
    std::vector<Big> bigs;
    // per event of a slot, in order: >= 0 an index into tasks, < 0 -(1 + index into bigs)
    std::vector<std::vector<int32_t>> slot_ev(ns);
    uint64_t bt = 0, rows = 0;
    for (uint32_t p = 0; p < ns; p++) {   // p: record slot; pair = p / R
        if (S.pre_status[p / R] != 0 || hdr[p].status != 0) continue;
        for (int e = 0; e < hdr[p].n_events; e++) {
            const AlnEvent &ev = events[(size_t)hdr[p].ev_off + e];
            if (ev.qs < 0) {
                slot_ev[p].push_back(-(int32_t)(1 + bigs.size()));
                bigs.push_back(Big{p, -1 - ev.qs, ev.qe, ev.ts, ev.te});
                continue;
            }
            NwTask t;
            t.out_idx = 0; t.pad = 0;
            t.pair = p; t.qs = ev.qs; t.ql = ev.qe - ev.qs + 1; t.ts = ev.ts; t.tl = ev.te - ev.ts + 1;
            t.cg_off = (uint32_t)(tasks.size() * ALN_CG_CAP); t.bt_off = bt; t.row_off = rows;
            bt += nw_bt_bytes(t.ql, t.tl);
            if (nw_class(t.ql, t.tl) == 2) rows += 11ull * t.ql;
            if (stats) { stats->dp_cells += (uint64_t)t.ql * t.tl; stats->algo_bytes += (uint64_t)(t.ql + t.tl + 3) / 4; }
            slot_ev[p].push_back((int32_t)tasks.size());
            tasks.push_back(t);
        }
    }
    // CIGAR runs of the events: nearly all have a handful; the first CG_HEAD runs of every task come back in one strided copy,
    // longer ones are fetched individually
    constexpr uint32_t CG_HEAD = 8;
    std::vector<uint32_t> cg_n(tasks.size()), cg_head(tasks.size() * (size_t)CG_HEAD);
    std::vector<std::vector<uint32_t>> cg_long(tasks.size());
    if (!tasks.empty()) {
        if (tasks.size() * (uint64_t)ALN_CG_CAP >= (1ull << 32)) return fsv_fail(ctx, FSV_EUNSUP, "too many DP events in one batch");
        TRY(run_nw(ctx, W, tasks, bt, rows, P));
        FSV_HIP(ctx, hipMemcpyAsync(cg_n.data(), W.cg_n.p, tasks.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        FSV_HIP(ctx, hipMemcpy2DAsync(cg_head.data(), CG_HEAD * 4, W.cg.p, (size_t)ALN_CG_CAP * 4, CG_HEAD * 4, tasks.size(), hipMemcpyDeviceToHost, ctx->stream));
        FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        bool any = false;
        for (size_t t = 0; t < tasks.size(); t++)
            if (cg_n[t] != 0xffffffffu && cg_n[t] > CG_HEAD) {
                cg_long[t].resize(cg_n[t]);
                FSV_HIP(ctx, hipMemcpyAsync(cg_long[t].data(), (const uint32_t *)W.cg.p + t * (size_t)ALN_CG_CAP, (size_t)cg_n[t] * 4, hipMemcpyDeviceToHost, ctx->stream));
                any = true;
            }
        if (any) FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    trace("dp");
    // oversize events
    std::vector<std::vector<uint32_t>> big_cg(bigs.size());
    if (!bigs.empty() && depth == 0) {
        // every box becomes a (query, target) pair of the second pass: its two sides are cut out of this pass's store
        if (!W.sub) W.sub = new AlnWs();
        AlnWs &W2 = *W.sub;
        const uint32_t nb = (uint32_t)bigs.size();
        PassIn S2;
        S2.n_refs = nb; S2.n_pairs = nb;
        S2.word_off.assign(2 * nb + 1, 0); S2.len.resize(2 * nb); S2.wper.resize(2 * nb); S2.thin.resize(2 * nb); S2.pair_t.resize(nb); S2.pre_status.assign(nb, 0);
        std::vector<BoxSrc> src(2 * nb);
        for (uint32_t b = 0; b < nb; b++) {
            const Big &B = bigs[b];
            const uint32_t rq = n_refs + B.slot / R, rt = S.pair_t[B.slot / R];
            S2.len[b] = B.te - B.ts + 1; S2.len[nb + b] = B.qe - B.qs + 1;
            src[b] = BoxSrc{word_off[rt], len[rt], 0, B.ts};
            src[nb + b] = BoxSrc{word_off[rq], len[rq], hdr[B.slot].rev, B.qs};
            S2.pair_t[b] = b;
            seed_window(P.w, (uint64_t)std::max(S2.len[b], S2.len[nb + b]), ALN_SUB_PER, S2.wper[b], S2.thin[b]);
            S2.wper[nb + b] = S2.wper[b]; S2.thin[nb + b] = S2.thin[b];
        }
        uint64_t w = 0;
        for (uint32_t r = 0; r < 2 * nb; r++) { S2.word_off[r] = (uint32_t)w; w += ((uint64_t)S2.len[r] + 15) / 16; }
        if (w + 8 >= (1ull << 32)) return fsv_fail(ctx, FSV_EUNSUP, "oversize events too large for one batch");
        S2.word_off[2 * nb] = (uint32_t)w;
        TRY(upload(ctx, W2.word_off, S2.word_off));
        TRY(upload(ctx, W2.len, S2.len));
        TRY(upload(ctx, W2.box_src, src));
        TRY(ensure(ctx, W2.store, (w + 8) * 4));
        FSV_HIP(ctx, hipMemsetAsync((uint32_t *)W2.store.p + w, 0, 32, ctx->stream));
        TRY(ensure(ctx, W2.nmask, (w + 8 + FSV_NM_LEAD) * 4));
        FSV_HIP(ctx, hipMemsetAsync((uint32_t *)W2.nmask.p + FSV_NM_LEAD + w, 0, 32, ctx->stream));
        hipLaunchKernelGGL(k_extract_boxes, dim3(fsv_grid_for(w, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)W.store.p, W.nm(), (const BoxSrc *)W2.box_src.p,
                           (const uint32_t *)W2.word_off.p, (const int32_t *)W2.len.p, 2 * nb, (uint32_t)w, (uint32_t *)W2.store.p, (uint32_t *)W2.nmask.p + FSV_NM_LEAD);
        FSV_HIP(ctx, hipGetLastError());
        PassOut O2;
        TRY(align_pass(ctx, W2, S2, P, 1, O2, nullptr));
        for (uint32_t b = 0; b < nb; b++) {
            if (O2.status[b] != 0) return fsv_fail(ctx, FSV_EINTERNAL, "a box of an oversize event came back without an alignment");
            big_cg[b].swap(O2.body[b]);
        }
        if (stats) stats->n_boxes += nb;
        trace("boxes");
    } else if (!bigs.empty()) {
        std::vector<CornerTask> ct(bigs.size());
        for (size_t b = 0; b < bigs.size(); b++) ct[b] = CornerTask{bigs[b].slot, bigs[b].qs, bigs[b].qe - bigs[b].qs + 1, bigs[b].ts, bigs[b].te - bigs[b].ts + 1};
        std::vector<int2> lr(bigs.size());
        TRY(upload(ctx, W.corner, ct));
        TRY(ensure(ctx, W.corner_out, bigs.size() * sizeof(int2)));
        hipLaunchKernelGGL(k_corner, dim3((uint32_t)bigs.size()), dim3(64), 0, ctx->stream, (const uint32_t *)W.store.p, W.nm(), (const uint32_t *)W.word_off.p,
                           (const int32_t *)W.len.p, (const uint32_t *)W.pair_q.p, (const uint32_t *)W.pair_t.p, (const AlnHeader *)W.hdr.p,
                           (const CornerTask *)W.corner.p, (int2 *)W.corner_out.p, P);
        FSV_HIP(ctx, hipGetLastError());
        FSV_HIP(ctx, hipMemcpyAsync(lr.data(), W.corner_out.p, bigs.size() * sizeof(int2), hipMemcpyDeviceToHost, ctx->stream));
        FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t b = 0; b < bigs.size(); b++) {
            const int ql = ct[b].ql, tl = ct[b].tl, l = lr[b].x, r = lr[b].y;
            push_cg(big_cg[b], 0, (uint32_t)l); push_cg(big_cg[b], 1, (uint32_t)(ql - l - r)); push_cg(big_cg[b], 2, (uint32_t)(tl - l - r)); push_cg(big_cg[b], 0, (uint32_t)r);
        }
        trace("corners");
    }
    if (stats) { stats->ms_dp = tdp.stop(); stats->n_events = tasks.size(); }
    // the ops of every record: M runs, events, M
    O.status.assign(ns, 0); O.body.assign(ns, std::vector<uint32_t>());
    for (uint32_t p = 0; p < ns; p++) {
        int32_t st = S.pre_status[p / R] != 0 ? S.pre_status[p / R] : hdr[p].status;
        if (st == 0) {
            const AlnHeader &h = hdr[p];
            std::vector<uint32_t> &c = O.body[p];
            int mstart = h.qbeg;
            for (int32_t e : slot_ev[p]) {
                if (e >= 0) {
                    const NwTask &T = tasks[(size_t)e];
                    push_cg(c, 0, (uint32_t)(T.qs - mstart));
                    if (cg_n[e] == 0xffffffffu) { st = FSV_ECAP; break; }
                    const uint32_t *runs = cg_n[e] > CG_HEAD ? cg_long[e].data() : cg_head.data() + (size_t)e * CG_HEAD;
                    for (uint32_t i = 0; i < cg_n[e]; i++) { uint32_t v = runs[i]; push_cg(c, v & 0xf, v >> 4); }
                    mstart = T.qs + T.ql;
                } else {
                    const Big &B = bigs[(size_t)(-e - 1)];
                    push_cg(c, 0, (uint32_t)(B.qs - mstart));
                    for (uint32_t v : big_cg[(size_t)(-e - 1)]) push_cg(c, v & 0xf, v >> 4);
                    mstart = B.qe + 1;
                }
            }
            if (st == 0) push_cg(c, 0, (uint32_t)(h.qend + 1 - mstart));
        }
        O.status[p] = st;
    }
    return FSV_OK;
}

} // namespace

extern "C" void fsv_aln_default_params(fsv_aln_params *P)
{
    if (!P) return;
    P->k = 19; P->w = 19; P->min_anchors = 3; P->lookback = 64; P->max_gap = 50000;
    P->a = 1; P->b = 19; P->q = 39; P->e = 3; P->q2 = 81; P->e2 = 1;
    P->pad = 24; P->max_mm_run = 4; P->xdrop = 100; P->max_cells = 1 << 26;
}

extern "C" int fsv_aln_last_stats(const fsv_ctx *ctx, fsv_aln_stats *out)
{
    if (!ctx || !out || !ctx->aln_ws) return FSV_EINVAL;
    *out = ((const AlnWs *)ctx->aln_ws)->stats;
    return FSV_OK;
}

static int check_aln_params(fsv_ctx *ctx, const fsv_aln_params &P)
{
    if (P.k < 1 || P.k > 31 || P.w < 1 || P.w > 255 || P.lookback != 64 || P.min_anchors < 2 || P.pad < 0 || P.a < 0 || P.b < 0 || P.q < 0 || P.e < 0 || P.max_cells < 1 || P.max_gap < 0 || P.xdrop < 0)
        return fsv_fail(ctx, FSV_EINVAL, "fsv_aln_params out of range");
    return FSV_OK;
}

static int nw_impl(fsv_ctx *ctx, const char *target, int32_t tl, const char *query, int32_t ql, const fsv_aln_params *params, int32_t *score,
                   uint32_t *cigar, uint32_t cigar_cap, uint32_t *n_cigar)
{
    if (!ctx || !target || !query || tl < 1 || ql < 1 || !score || !cigar || !n_cigar) return FSV_EINVAL;
    fsv_aln_params P;
    if (params) P = *params; else fsv_aln_default_params(&P);
    TRY(check_aln_params(ctx, P));
    if ((int64_t)tl * ql > P.max_cells) return fsv_fail(ctx, FSV_EUNSUP, "event larger than max_cells");
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    AlnWs &W = *aln_ws_get(ctx);
    std::vector<uint32_t> word_off; std::vector<int32_t> len;
    TRY(pack_pairs(ctx, W, {query, target}, {(uint64_t)ql, (uint64_t)tl}, word_off, len));
    std::vector<AlnHeader> hdr(1); memset(&hdr[0], 0, sizeof(AlnHeader));
    TRY(upload(ctx, W.hdr, hdr));
    TRY(upload(ctx, W.pair_q, std::vector<uint32_t>{0u}));
    TRY(upload(ctx, W.pair_t, std::vector<uint32_t>{1u}));
    std::vector<NwTask> tasks(1);
    tasks[0] = NwTask{0u, 0, ql, 0, tl, 0u, 0ull, 0ull, 0u, 0u};
    TRY(run_nw(ctx, W, tasks, nw_bt_bytes(ql, tl), nw_class(ql, tl) == 2 ? 11ull * ql : 0, P));
    uint32_t n = 0; int32_t sc = 0;
    FSV_HIP(ctx, hipMemcpyAsync(&n, W.cg_n.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    FSV_HIP(ctx, hipMemcpyAsync(&sc, W.scores.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n == 0xffffffffu || n > cigar_cap) return FSV_ECAP;
    FSV_HIP(ctx, hipMemcpyAsync(cigar, W.cg.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_cigar = n; *score = sc;
    return FSV_OK;
}

static int align_batch_impl(fsv_ctx *ctx, const char *contig_seq, const uint64_t *contig_off, uint32_t n_contigs, const uint32_t *contig_ref,
                            const char *ref_seq, const uint64_t *ref_off, uint32_t n_refs, const fsv_aln_params *params, fsv_alns *out)
{
    if (!ctx || !out || !out->rec || !out->cigar || !out->contig_status) return FSV_EINVAL;
    out->n_rec = 0; out->n_cigar = 0;
    if (n_contigs == 0) return FSV_OK;
    // contig_seq == NULL: align the contigs of the last fsv_assemble_batch on this context straight from device memory
    const bool from_dev = contig_seq == nullptr;
    if (from_dev) {
        if (!ctx->last_contigs_dev || ctx->last_contig_off.size() != (size_t)n_contigs + 1) return fsv_fail(ctx, FSV_EINVAL, "no assembled contigs of that count on this context");
        contig_off = ctx->last_contig_off.data();
    }
    if (!contig_off || !contig_ref || !ref_seq || !ref_off) return FSV_EINVAL;
    if (out->rec_cap < n_contigs) return fsv_fail(ctx, FSV_ECAP, "rec_cap must be >= n_contigs");
    fsv_aln_params P;
    if (params) P = *params; else fsv_aln_default_params(&P);
    TRY(check_aln_params(ctx, P));
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    AlnWs &W = *aln_ws_get(ctx);
    memset(&W.stats, 0, sizeof(W.stats));
    Timer ttot(ctx);
    const uint32_t np = n_contigs, nr = n_refs + n_contigs;
    // sequences in the store: the reference windows first, then the contigs; every window is sketched once with
    // w = max(P.w, L/3000 + 1), L = the longest sequence of its group, and its contigs use the same w
    PassIn S;
    S.n_refs = n_refs; S.n_pairs = np;
    std::vector<const char *> seq(nr); std::vector<uint64_t> slen(nr);
    S.wper.resize(nr); S.thin.assign(nr, 1); S.pair_t.resize(np); S.pre_status.assign(np, 0);
    std::vector<uint64_t> group_len(n_refs, 0);
    for (uint32_t r = 0; r < n_refs; r++) {
        seq[r] = ref_seq + ref_off[r]; slen[r] = ref_off[r + 1] - ref_off[r];
        if (slen[r] == 0) return fsv_fail(ctx, FSV_EINVAL, "empty reference window");
        group_len[r] = slen[r];
    }
    for (uint32_t p = 0; p < np; p++) {
        if (contig_ref[p] >= n_refs) return fsv_fail(ctx, FSV_EINVAL, "contig_ref out of range");
        seq[n_refs + p] = from_dev ? nullptr : contig_seq + contig_off[p]; slen[n_refs + p] = contig_off[p + 1] - contig_off[p];
        if (slen[n_refs + p] == 0) return fsv_fail(ctx, FSV_EINVAL, "empty contig");
        group_len[contig_ref[p]] = std::max(group_len[contig_ref[p]], slen[n_refs + p]);
        S.pair_t[p] = contig_ref[p];
    }
    // seeds of long windows: w grows with the longest sequence of the group; beyond w = 255 (~760 kb: the whole-genome BED has a
    // 1.15 Mb region) the minimizers are thinned by hash instead (k_thin_seeds), so any window below 2^23 bases is aligned
    for (uint32_t r = 0; r < n_refs; r++) seed_window(P.w, group_len[r], 3000, S.wper[r], S.thin[r]);
    for (uint32_t p = 0; p < np; p++) {
        const uint32_t r = contig_ref[p];
        const uint64_t L = group_len[r];
        S.wper[n_refs + p] = S.wper[r]; S.thin[n_refs + p] = S.thin[r];
        if (L >= (1u << 23) || (!(P.k & 1) && L / 3000 + 1 > 64)) S.pre_status[p] = FSV_EUNSUP; // the replay kernel (even k) holds w <= 64
        else if (slen[n_refs + p] < (uint64_t)P.k || slen[r] < (uint64_t)P.k) S.pre_status[p] = 1;
    }
    TRY(pack_pairs(ctx, W, seq, slen, S.word_off, S.len, from_dev ? ctx->last_contigs_dev + contig_off[0] : nullptr, n_refs));
    PassOut O;
    TRY(align_pass(ctx, W, S, P, 0, O, &W.stats));
    const uint32_t R = O.R;
    const std::vector<AlnHeader> &hdr = O.hdr;
    const std::vector<int32_t> &len = S.len;
    W.stats.n_pairs = np;
    // stitch: S, the record's ops, S -- one record per used slot, the primary first
    struct Rec { uint32_t contig, slot; std::vector<uint32_t> c; };
    std::vector<Rec> recs;
    std::vector<GapQuery> gaps;
    std::vector<int32_t> status(np, 0);
    for (uint32_t cp = 0; cp < np; cp++) {
        const uint32_t rq = n_refs + cp;
        W.stats.algo_bytes += (uint64_t)(len[rq] + len[S.pair_t[cp]] + 3) / 4;
        int32_t st = O.status[cp * R];
        const size_t first_rec = recs.size(), first_gap = gaps.size();
        for (uint32_t r = 0; r < R && st == 0; r++) {
            const uint32_t p = cp * R + r;
            const AlnHeader &h = hdr[p];
            if (O.status[p] != 0) { if (O.status[p] != 1) st = O.status[p]; break; }   // slots fill up in order; 1 = an empty slot
            std::vector<uint32_t> c;
            push_cg(c, 4, (uint32_t)h.qbeg);
            for (uint32_t v : O.body[p]) push_cg(c, v & 0xf, v >> 4);
            push_cg(c, 4, (uint32_t)(len[rq] - 1 - h.qend));
            // gaps flanked by M on both sides: how far could each travel left (answered on the device below)
            int toff = h.tbeg, qoff = 0;
            for (size_t k = 0; k < c.size(); k++) {
                const uint32_t op = c[k] & 0xf; const int l = (int)(c[k] >> 4);
                if (op == 0) { toff += l; qoff += l; }
                else if (op == 4) qoff += l;
                else {
                    if (k > 0 && k + 1 < c.size() && (c[k - 1] & 0xf) == 0 && (c[k + 1] & 0xf) == 0)
                        gaps.push_back(GapQuery{p, op == 1, op == 1 ? qoff : toff, l, std::min(toff - h.tbeg, qoff - h.qbeg)});
                    if (op == 2) toff += l; else qoff += l;
                }
            }
            recs.push_back(Rec{cp, p, std::move(c)});
        }
        if (st != 0) { recs.resize(first_rec); gaps.resize(first_gap); }
        status[cp] = st;
    }
    std::vector<int32_t> max_shift(gaps.size());
    if (!gaps.empty()) {
        TRY(upload(ctx, W.gaps, gaps));
        TRY(ensure(ctx, W.gap_shift, gaps.size() * 4));
        hipLaunchKernelGGL(k_gap_shift, dim3((uint32_t)gaps.size()), dim3(64), 0, ctx->stream, (const uint32_t *)W.store.p, W.nm(), (const uint32_t *)W.word_off.p,
                           (const int32_t *)W.len.p, (const uint32_t *)W.pair_q.p, (const uint32_t *)W.pair_t.p, (const AlnHeader *)W.hdr.p,
                           (const GapQuery *)W.gaps.p, (int32_t *)W.gap_shift.p);
        FSV_HIP(ctx, hipGetLastError());
        FSV_HIP(ctx, hipMemcpyAsync(max_shift.data(), W.gap_shift.p, gaps.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        FSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    {
        size_t gi = 0;
        for (Rec &rc : recs) {
            std::vector<uint32_t> &c = rc.c;
            bool shrink = false;
            for (size_t k = 1; k + 1 < c.size(); k++) {
                const uint32_t op = c[k] & 0xf;
                if ((op != 1 && op != 2) || (c[k - 1] & 0xf) != 0 || (c[k + 1] & 0xf) != 0) continue;
                const uint32_t prev = c[k - 1] >> 4, l = std::min<uint32_t>(prev, (uint32_t)max_shift[gi++]);   // same gaps, same order as collected
                c[k - 1] -= l << 4; c[k + 1] += l << 4;
                shrink |= l == prev;
            }
            if (shrink) {
                std::vector<uint32_t> d;
                for (uint32_t v : c) push_cg(d, v & 0xf, v >> 4);
                c.swap(d);
            }
            const AlnHeader &h = hdr[rc.slot];
            if (out->n_rec >= out->rec_cap) return fsv_fail(ctx, FSV_ECAP, "record buffer too small (rec_cap >= 5 x n_contigs holds every case)");
            if (out->n_cigar + c.size() > out->cigar_cap) return fsv_fail(ctx, FSV_ECAP, "cigar buffer too small");
            fsv_aln_rec &rr = out->rec[out->n_rec++];
            rr.ref_start = h.tbeg; rr.ref_end = h.tend + 1; rr.q_start = h.qbeg; rr.q_end = h.qend + 1; rr.n_cigar = (uint32_t)c.size();
            rr.n_chain = (uint32_t)h.n_chain; rr.cigar_off = out->n_cigar; rr.contig = rc.contig; rr.rev = (uint8_t)h.rev; rr.mapq = 60; rr.pad[0] = rr.pad[1] = 0;
            memcpy(out->cigar + out->n_cigar, c.data(), c.size() * 4);
            out->n_cigar += c.size();
            W.stats.algo_bytes += c.size() * 4;
        }
    }
    for (uint32_t cp = 0; cp < np; cp++) out->contig_status[cp] = status[cp];
    W.stats.ms_total = ttot.stop();
    return FSV_OK;
}

extern "C" int fsv_nw(fsv_ctx *ctx, const char *target, int32_t tl, const char *query, int32_t ql, const fsv_aln_params *params, int32_t *score,
                      uint32_t *cigar, uint32_t cigar_cap, uint32_t *n_cigar)
{
    FSV_GUARD(ctx, nw_impl(ctx, target, tl, query, ql, params, score, cigar, cigar_cap, n_cigar));
}

extern "C" int fsv_align_batch(fsv_ctx *ctx, const char *contig_seq, const uint64_t *contig_off, uint32_t n_contigs, const uint32_t *contig_ref,
                               const char *ref_seq, const uint64_t *ref_off, uint32_t n_refs, const fsv_aln_params *params, fsv_alns *out)
{
    FSV_GUARD(ctx, align_batch_impl(ctx, contig_seq, contig_off, n_contigs, contig_ref, ref_seq, ref_off, n_refs, params, out));
}
