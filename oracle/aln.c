/* oracle/aln.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the contig-vs-reference alignment FocalSV obtains from
 *   minimap2 -a -x asm5 --cs -r2k ref_chr.fa assemblies.fa | samtools sort
 * (focalsv/4_sv_calling/Dippav/DipPAV_variant_call.py:103-108), as consumed by
 * extract_contig_signature_CCS.py:14-47, 342-432 (reference_name, pos, reference_end, cigar, is_reverse, mapq).
 *
 * minimap2 2.24 itself is NOT under /root/reference (conda pin requirement.yaml:12), so this part of the
 * oracle is "parity unpinned" against minimap2.  What is pinned:
 *   - the DP recurrence, tie-breaking and backtrack state machine follow the in-tree ksw2
 *     (software/hifiasm-0.14/ksw2_extz2_sse.c:23-305 left-aligned branch :171-196; ksw2.h:115-150 ksw_backtrack),
 *     checked in single-affine mode against tests/golden/ksw_extz2.json minted from that code;
 *   - minimizer seeds are ha_sketch without HPC (sketch.cpp:39-137), k = 19 as in minimap2's asm5 preset;
 *   - scoring is the published asm5 preset: A=1 B=19 O=39,81 E=3,1.
 * Pipeline: seeds unique in both sequences -> co-linear chains (look-back 64) on the majority strand, one chain of the
 * other strand (an inverted piece: a record of its own, the record around it is cut) ->
 * gap-free runs between anchors stay 'M'; everything else is merged into events, padded by up to
 * ALN_PAD identical bases per side (so that a gap can be left-aligned past the seed boundary) and
 * aligned globally with dual-affine gaps -- an event of more than max_cells cells (both copies of a duplication between two
 * unique seeds) is seeded again on its own and walked the same way (sub_align), what is still too large inside it is closed
 * from its corners (corner_event): no contig is refused; contig ends are extended gap-free with an X-drop and the
 * rest is soft-clipped; last, every gap of the stitched CIGAR is moved to its leftmost position (shift_gaps_left).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static char comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; } }
/* A reference window may hold N (anything that is not A/C/G/T).  An N pairs with nothing -- no gap-free run, no padding, no exact
 * match across it -- and costs ORC_SC_AMBI in a score, as in minimap2 (sc_ambi = 1: the alignment runs through a short run of N
 * as 'M').  For SEEDING an N holds a base hashed from its position in the window (n_substitute: the HIP path's 2-bit store has to
 * hold something): the k-mers of an N run then look like random sequence, unique and matching nothing, where minimap2 skips
 * k-mers with an N; a k-mer with one or two N can still seed where the contig happens to carry the hashed base -- on the true diagonal. */
#define ORC_SC_AMBI 1
static inline int is_acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }
static inline int pair_score(char qc, char tc, const orc_aln_params *P) { return !is_acgt(tc) ? -ORC_SC_AMBI : (qc == tc ? P->a : -P->b); }
static inline uint32_t n_substitute(uint32_t pos)
{
    uint32_t x = pos * 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13;
    return x >> 30;
}
/* the window of the alignment in progress and its seeding copy (NULL: no N in it): sub_align seeds a box from the copy */
static __thread const char *g_ref = NULL, *g_ref_seed = NULL;
static __thread int g_ref_len = 0;

void orc_aln_default_params(orc_aln_params *P)
{
    P->k = 19; P->w = 19; P->min_anchors = 3; P->lookback = 64; P->max_gap = 50000;
    P->a = 1; P->b = 19; P->q = 39; P->e = 3; P->q2 = 81; P->e2 = 1;
    P->pad = 24; P->max_mm_run = 4; P->xdrop = 100; P->max_cells = 1 << 26;
}

/* ------------------------------------------------------------------ global dual-affine DP (ksw2 conventions)
 * i indexes the target (reference), j the query.  Per cell one byte:
 *   bits 0-2  which state gives H: 0 diagonal, 1 E (deletion), 2 F (insertion), 3 E2, 4 F2 (strictly greater wins, in that order)
 *   0x08 E continues, 0x10 F continues, 0x20 E2 continues, 0x40 F2 continues (strictly better than opening)
 * q2 < 0 disables the second gap model (single affine, as ksw_extz2_sse). */
#define NEG (-(1 << 29))
int orc_nw(const char *t, int tl, const char *q, int ql, const orc_aln_params *P, uint32_t *cigar, int cigar_cap, int *n_cigar, uint8_t *bt)
{
    const int two = P->q2 >= 0;
    int32_t *H = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ql + 1) * 2), *Hn = H + (ql + 1), *tmp;
    int32_t *base = H;
    int i, j, n = 0, state = 0, score;
    /* row -1 */
    H[0] = 0;
    for (j = 1; j <= ql; j++) {
        int g1 = -(P->q + P->e * j), g2 = two ? -(P->q2 + P->e2 * j) : NEG;
        H[j] = g1 > g2 ? g1 : g2;
    }
    {
        int32_t *E = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ql + 1) * 2), *E2 = E + (ql + 1);
        for (j = 0; j <= ql; j++) { E[j] = NEG; E2[j] = NEG; }
        /* E[j] = value of the E state entering cell (i, j) from (i-1, j): initialised from the boundary row */
        for (j = 1; j <= ql; j++) {
            /* coming down from H(-1, j-1 .. ) : E(0,j) = H(-1,j) - q - e */
            E[j] = H[j] - P->q - P->e;
            E2[j] = two ? H[j] - P->q2 - P->e2 : NEG;
        }
        for (i = 0; i < tl; i++) {
            int32_t hleft, f, f2, hdiag;
            {
                int g1 = -(P->q + P->e * (i + 1)), g2 = two ? -(P->q2 + P->e2 * (i + 1)) : NEG;
                Hn[0] = g1 > g2 ? g1 : g2; /* H(i, -1) */
            }
            hleft = Hn[0];
            f = hleft - P->q - P->e; f2 = two ? hleft - P->q2 - P->e2 : NEG;
            hdiag = H[0]; /* H(i-1, -1) */
            for (j = 1; j <= ql; j++) {
                int32_t z = hdiag + pair_score(q[j - 1], t[i], P);
                int32_t a = E[j], b = f, a2 = E2[j], b2 = f2, h, o;
                uint8_t d = 0;
                h = z;
                if (a > h) { h = a; d = 1; }
                if (b > h) { h = b; d = 2; }
                if (two && a2 > h) { h = a2; d = 3; }
                if (two && b2 > h) { h = b2; d = 4; }
                /* next-cell gap states and their continuation flags */
                o = h - P->q;
                if (a > o) { d |= 0x08; E[j] = a - P->e; } else E[j] = o - P->e;
                if (b > o) { d |= 0x10; f = b - P->e; } else f = o - P->e;
                if (two) {
                    o = h - P->q2;
                    if (a2 > o) { d |= 0x20; E2[j] = a2 - P->e2; } else E2[j] = o - P->e2;
                    if (b2 > o) { d |= 0x40; f2 = b2 - P->e2; } else f2 = o - P->e2;
                }
                bt[(size_t)i * ql + (j - 1)] = d;
                hdiag = H[j];
                Hn[j] = h;
            }
            tmp = H; H = Hn; Hn = tmp;
        }
        free(E);
    }
    score = H[ql];
    /* backtrack: ksw_backtrack (ksw2.h:119-150), emitted end-to-start then reversed */
    i = tl - 1; j = ql - 1;
    while (i >= 0 && j >= 0) {
        uint8_t d = bt[(size_t)i * ql + j];
        int op;
        if (state == 0) state = d & 7;
        else if (!((d >> (state + 2)) & 1)) state = 0;
        if (state == 0) state = d & 7;
        if (state == 0) { op = 0; i--; j--; }
        else if (state == 1 || state == 3) { op = 2; i--; }
        else { op = 1; j--; }
        if (n && (cigar[n - 1] & 0xf) == (uint32_t)op) cigar[n - 1] += 1u << 4;
        else { if (n == cigar_cap) { free(base); return NEG; } cigar[n++] = 1u << 4 | (uint32_t)op; }
    }
    if (i >= 0) { if (n && (cigar[n - 1] & 0xf) == 2) cigar[n - 1] += (uint32_t)(i + 1) << 4; else { if (n == cigar_cap) { free(base); return NEG; } cigar[n++] = (uint32_t)(i + 1) << 4 | 2; } }
    if (j >= 0) { if (n && (cigar[n - 1] & 0xf) == 1) cigar[n - 1] += (uint32_t)(j + 1) << 4; else { if (n == cigar_cap) { free(base); return NEG; } cigar[n++] = (uint32_t)(j + 1) << 4 | 1; } }
    for (i = 0; i < n / 2; i++) { uint32_t x = cigar[i]; cigar[i] = cigar[n - 1 - i]; cigar[n - 1 - i] = x; }
    *n_cigar = n;
    free(base);
    return score;
}

/* ------------------------------------------------------------------ seeds + chain */
typedef struct { int32_t qe, te; } anc_t;
static int anc_cmp(const void *a, const void *b)
{
    const anc_t *x = (const anc_t *)a, *y = (const anc_t *)b;
    if (x->qe != y->qe) return x->qe < y->qe ? -1 : 1;
    if (x->te != y->te) return x->te < y->te ? -1 : 1;
    return 0;
}

static inline int ilog2_32(uint32_t v) { int l = 0; while (v >>= 1) l++; return l; }

/* co-linear chain DP over anchors sorted by (qe, te): look back P->lookback anchors, the best-scoring predecessor, the nearer
 * one on ties; f / pre are caller-owned arrays of n entries.  Returns the index of the best chain end (the first on ties). */
static int chain_dp(const anc_t *a, int n, const orc_aln_params *P, int32_t *f, int32_t *pre)
{
    int i, j, best = -1;
    for (i = 0; i < n; i++) {
        int32_t bs = P->k, bp = -1;
        int lo = i - P->lookback < 0 ? 0 : i - P->lookback;
        for (j = i - 1; j >= lo; j--) {
            int32_t dq = a[i].qe - a[j].qe, dt = a[i].te - a[j].te, gap, sc;
            if (dq <= 0 || dt <= 0) continue;
            gap = dq > dt ? dq - dt : dt - dq;
            if (gap > P->max_gap) continue;
            sc = dq < dt ? dq : dt;
            if (sc > P->k) sc = P->k;
            if (gap) sc -= (gap >> 7) + (ilog2_32((uint32_t)gap) >> 1) + 1;
            sc += f[j];
            if (sc > bs) { bs = sc; bp = j; }
        }
        f[i] = bs; pre[i] = bp;
    }
    for (i = 0; i < n; i++) if (best < 0 || f[i] > f[best]) best = i;
    return best;
}

/* chains of (contig, reference); a chain's contig coordinates are those of the contig in the chain's own orientation (on the
 * reverse strand: of its reverse complement).
 * The best chain on the contig's majority strand is the primary alignment.  Like minimap2, which reports what the primary
 * leaves uncovered as supplementary alignments (DipPAV calls SVs beyond the chaining gap from consecutive records of one
 * contig, extract_contig_signature_CCS.py:251-327), the anchors inside the query interval of a chain are then taken out and
 * the rest is chained again, up to ORC_ALN_MAJ_REC chains on the majority strand; a supplementary chain needs a score of
 * ORC_ALN_SUP_MIN.
 * Then the anchors of the OTHER strand are chained once: an inverted piece of the contig.  minimap2 reports it as a
 * reverse-strand supplementary record and breaks the alignment around it (z-drop with the inversion test), and DipPAV's split
 * rule only pairs consecutive records of the same strand (extract_contig_signature_CCS.py:286) -- so an inversion yields no
 * INS / DEL call.  Here: when that chain's span on the contig lies strictly between the k-mers of two consecutive anchors
 * of a majority-strand chain, and both parts keep min_anchors anchors, the majority chain is cut between the two (the part
 * behind the cut becomes one more record); the minority chain itself is the last record.
 * chain_n[r] anchors of chain r follow each other in cq/ct, chain_rev[r] is its strand.  Returns the number of chains. */
int orc_aln_chains(const orc_mz *mq, int nq, int lenq, const orc_mz *mt, int nt, const orc_aln_params *P, int *rev_out,
                   int32_t *cq, int32_t *ct, int cap, int *chain_n, uint8_t *chain_rev, int max_rec)
{
    int i = 0, j = 0, n = 0, nb = 0, nf = 0, nr = 0, rev, c, n_rec = 0, used = 0, maj_rec;
    int lim = nq < nt ? nq : nt;
    anc_t *a, *b; uint8_t *sr; int32_t *f, *pre;
    if (lim <= 0) return 0;
    a = (anc_t *)malloc(sizeof(anc_t) * (size_t)lim * 2); b = a + lim;
    sr = (uint8_t *)malloc((size_t)lim);
    while (i < nq && j < nt) {
        if (mq[i].hash < mt[j].hash) i++;
        else if (mq[i].hash > mt[j].hash) j++;
        else {
            sr[n] = mq[i].rev ^ mt[j].rev;
            a[n].qe = sr[n] ? (lenq - 1) - ((int32_t)mq[i].pos - mq[i].span + 1) : (int32_t)mq[i].pos;
            a[n].te = (int32_t)mt[j].pos;
            if (sr[n]) nr++; else nf++;
            n++; i++; j++;
        }
    }
    rev = nr > nf;
    for (i = 0, j = 0; i < n; i++) if (sr[i] == rev) a[j++] = a[i]; else b[nb++] = a[i];
    n = j;
    free(sr);
    *rev_out = rev;
    if (n < P->min_anchors) { free(a); return 0; }
    qsort(a, (size_t)n, sizeof(anc_t), anc_cmp);
    qsort(b, (size_t)nb, sizeof(anc_t), anc_cmp);
    f = (int32_t *)malloc(sizeof(int32_t) * (size_t)lim * 2); pre = f + lim;
    maj_rec = max_rec < ORC_ALN_MAJ_REC ? max_rec : ORC_ALN_MAJ_REC;
    while (n_rec < maj_rec && n >= P->min_anchors) {
        int best, cnt = 0, qlo, qhi;
        best = chain_dp(a, n, P, f, pre);
        for (c = best; c >= 0; c = pre[c]) cnt++;
        if (cnt < P->min_anchors || used + cnt > cap || (n_rec > 0 && f[best] < ORC_ALN_SUP_MIN)) break;
        { int k2 = used + cnt; for (c = best; c >= 0; c = pre[c]) { k2--; cq[k2] = a[c].qe; ct[k2] = a[c].te; } }
        chain_rev[n_rec] = (uint8_t)rev;
        chain_n[n_rec++] = cnt;
        qlo = cq[used]; qhi = cq[used + cnt - 1];
        used += cnt;
        for (i = 0, j = 0; i < n; i++) if (a[i].qe < qlo || a[i].qe > qhi) a[j++] = a[i]; /* the rest, still in query order */
        n = j;
    }
    if (n_rec > 0 && n_rec < max_rec && nb >= P->min_anchors) {
        int best = chain_dp(b, nb, P, f, pre), cnt = 0;
        for (c = best; c >= 0; c = pre[c]) cnt++;
        if (cnt >= P->min_anchors && used + cnt <= cap && f[best] >= ORC_ALN_SUP_MIN) {
            int first = best, lo_a, hi_a, r, off = 0, cut_r = -1, cut_s = -1;
            for (c = best; c >= 0; c = pre[c]) first = c;
            /* span of the chain's k-mers on the contig, in the majority strand's coordinates */
            lo_a = (lenq - 1) - b[best].qe; hi_a = (lenq - 1) - (b[first].qe - P->k + 1);
            for (r = 0; r < n_rec && cut_r < 0; off += chain_n[r], r++) {
                int s;
                for (s = P->min_anchors - 1; s + 1 + P->min_anchors <= chain_n[r]; s++)
                    if (cq[off + s] < lo_a && hi_a < cq[off + s + 1] - P->k + 1) { cut_r = r; cut_s = s; break; }
            }
            if (cut_r >= 0 && n_rec + 1 < max_rec) {
                /* the part behind the cut moves to the end of the majority chains' anchors and becomes the next record */
                int o2 = 0, tail;
                int32_t *tq, *tt;
                for (r = 0; r < cut_r; r++) o2 += chain_n[r];
                tail = chain_n[cut_r] - (cut_s + 1);
                tq = (int32_t *)malloc(sizeof(int32_t) * (size_t)tail * 2); tt = tq + tail;
                memcpy(tq, cq + o2 + cut_s + 1, sizeof(int32_t) * (size_t)tail); memcpy(tt, ct + o2 + cut_s + 1, sizeof(int32_t) * (size_t)tail);
                memmove(cq + o2 + cut_s + 1, cq + o2 + chain_n[cut_r], sizeof(int32_t) * (size_t)(used - o2 - chain_n[cut_r]));
                memmove(ct + o2 + cut_s + 1, ct + o2 + chain_n[cut_r], sizeof(int32_t) * (size_t)(used - o2 - chain_n[cut_r]));
                memcpy(cq + used - tail, tq, sizeof(int32_t) * (size_t)tail); memcpy(ct + used - tail, tt, sizeof(int32_t) * (size_t)tail);
                free(tq);
                chain_n[cut_r] = cut_s + 1;
                chain_rev[n_rec] = (uint8_t)rev;
                chain_n[n_rec++] = tail;
            }
            { int k2 = used + cnt; for (c = best; c >= 0; c = pre[c]) { k2--; cq[k2] = b[c].qe; ct[k2] = b[c].te; } }
            chain_rev[n_rec] = (uint8_t)!rev;
            chain_n[n_rec++] = cnt;
            used += cnt;
        }
    }
    free(f); free(a);
    return n_rec;
}

int orc_aln_chain(const orc_mz *mq, int nq, int lenq, const orc_mz *mt, int nt, const orc_aln_params *P, int *rev_out,
                  int32_t *cq, int32_t *ct, int cap)
{
    int cn[1] = {0}; uint8_t cr[1] = {0};
    return orc_aln_chains(mq, nq, lenq, mt, nt, P, rev_out, cq, ct, cap, cn, cr, 1) ? cn[0] : 0;
}

/* ------------------------------------------------------------------ one contig against one reference window */
static void push(uint32_t *cg, int *n, int cap, uint32_t op, uint32_t len)
{
    if (!len) return;
    if (*n && (cg[*n - 1] & 0xf) == op) { cg[*n - 1] += len << 4; return; }
    if (*n < cap) cg[(*n)++] = len << 4 | op;
}


/* Indel left alignment over the finished CIGAR, the rule of minimap2's mm_fix_cigar (align.c; minimap2 2.24 is not under
 * /root/reference -- requirement.yaml:12 -- so this restates its published behaviour: "for each I/D flanked by M on both
 * sides, move it left while the base entering the gap on the left equals the base leaving it on the right", bounded by the
 * preceding M run; an M run shifted away completely leaves two neighbouring gap ops, same-op neighbours are merged).
 * A deletion compares reference bases only, an insertion query bases only, so the alignment score is unchanged.
 * On the query anything that is not A/C/G/T compares as 'A' (the device store has two bits per base; contigs hold none); on the
 * reference an N equals an N only (ref_gap_max_shift). */
static inline char acgt(char c) { return (c == 'C' || c == 'G' || c == 'T') ? c : 'A'; }
int orc_gap_max_shift(const char *s, int off, int len, int cap)
{
    int l = 0;
    while (l < cap && acgt(s[off - 1 - l]) == acgt(s[off + len - 1 - l])) l++;
    return l;
}
static inline char ref_norm(char c) { return is_acgt(c) ? c : 'N'; }
static int ref_gap_max_shift(const char *s, int off, int len, int cap)      /* the reference side: an N equals an N only */
{
    int l = 0;
    while (l < cap && ref_norm(s[off - 1 - l]) == ref_norm(s[off + len - 1 - l])) l++;
    return l;
}
static void shift_gaps_left(uint32_t *cg, int *n_io, const char *Q, const char *ref, int tbeg)
{
    int n = *n_io, k, toff = tbeg, qoff = 0, m = 0;
    for (k = 0; k < n; k++) {
        uint32_t op = cg[k] & 0xf, len = cg[k] >> 4;
        if (op == 0) { toff += (int)len; qoff += (int)len; }
        else if (op == 4) qoff += (int)len;
        else {
            if (k > 0 && k < n - 1 && (cg[k - 1] & 0xf) == 0 && (cg[k + 1] & 0xf) == 0) {
                int prev = (int)(cg[k - 1] >> 4);
                int l = op == 1 ? orc_gap_max_shift(Q, qoff, (int)len, prev) : ref_gap_max_shift(ref, toff, (int)len, prev);
                if (l > 0) { cg[k - 1] -= (uint32_t)l << 4; cg[k + 1] += (uint32_t)l << 4; toff -= l; qoff -= l; }
            }
            if (op == 2) toff += (int)len; else qoff += (int)len;
        }
    }
    for (k = 0; k < n; k++) {
        if ((cg[k] >> 4) == 0) continue;
        if (m && (cg[m - 1] & 0xf) == (cg[k] & 0xf)) cg[m - 1] += cg[k] & ~0xfu;
        else cg[m++] = cg[k];
    }
    *n_io = m;
}

/* ------------------------------------------------------------------ events larger than max_cells
 * With seeds that are unique in both sequences, both copies of a duplicated stretch fall between two anchors: a 6 kb tandem
 * duplication makes a 6 k x 12 k event.  minimap2 has anchors inside the copies (its seeds need not be unique) and aligns
 * only the seam between them.  Restated here in two steps:
 *   1. the box between the two anchors is seeded again on its own -- minimizers that occur at most ORC_ALN_SUB_OCC times in
 *      each side of the box, every pair of equal hash an anchor, same strand only -- chained (one chain, the same DP) and
 *      walked like a contig, but end to end: the box's first and last base pairs are fixed (the outer anchors);
 *   2. an event of that inner walk that is still larger than max_cells is not aligned base by base: gap-free X-drop
 *      extensions from its two corners, the remainder one insertion + one deletion (what a global alignment of unrelated
 *      sequences comes to under asm5's long-gap model).
 * So no event is ever refused. */
static void corner_event(const char *Qb, int ql, const char *Tb, int tl, const orc_aln_params *P, uint32_t *cg, int *n, int cap)
{
    int lim = ql < tl ? ql : tl, i, x = 0, best = 0, l = 0, r = 0;
    for (i = 0; i < lim; i++) {
        x += pair_score(Qb[i], Tb[i], P);
        if (x > best) { best = x; l = i + 1; }
        if (best - x > P->xdrop) break;
    }
    x = 0; best = 0;
    for (i = 0; i < lim - l; i++) {
        x += pair_score(Qb[ql - 1 - i], Tb[tl - 1 - i], P);
        if (x > best) { best = x; r = i + 1; }
        if (best - x > P->xdrop) break;
    }
    push(cg, n, cap, 0, (uint32_t)l);
    push(cg, n, cap, 1, (uint32_t)(ql - l - r));
    push(cg, n, cap, 2, (uint32_t)(tl - l - r));
    push(cg, n, cap, 0, (uint32_t)r);
}

static int mz_hash_pos_cmp(const void *a, const void *b)
{
    const orc_mz *x = (const orc_mz *)a, *y = (const orc_mz *)b;
    if (x->hash != y->hash) return x->hash < y->hash ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return 0;
}
/* minimizers whose hash occurs at most max_occ times in the list, sorted by (hash, position) */
int orc_occ_sorted(orc_mz *mz, int n, int max_occ)
{
    int i, j, k, m = 0;
    qsort(mz, (size_t)n, sizeof(orc_mz), mz_hash_pos_cmp);
    for (i = 0; i < n; i = j) {
        for (j = i + 1; j < n && mz[j].hash == mz[i].hash; j++) {}
        if (j - i <= max_occ) for (k = i; k < j; k++) mz[m++] = mz[k];
    }
    return m;
}

static void walk(const char *Q, const char *ref, const int32_t *cq, const int32_t *ct, int nch, int qbeg, int qend, int tend,
                 const orc_aln_params *P, uint32_t *cigar, int *n_io, int cigar_cap, int depth);

/* seed window of a sequence pair whose longer side has L bases; *thin > 1: every thin-th minimizer by hash survives */
static int seed_window(int w0, int L, int per, int *thin)
{
    int w = w0;
    *thin = 1;
    if (L / per + 1 > w) w = L / per + 1;
    if (w > 255) { *thin = (w + 254) / 255; w = 255; }
    return w;
}
static int sketch_thinned(const char *s, int len, int w, int k, int thin, orc_mz *mz, int cap)
{
    int n = orc_sketch(s, len, w, k, 0, mz, cap), i, j = 0;
    if (thin > 1) { for (i = 0; i < n; i++) if ((mz[i].hash >> 11) % (uint64_t)thin == 0) mz[j++] = mz[i]; n = j; }
    return n;
}

static void sub_align(const char *Qb, int ql, const char *Tb, int tl, const orc_aln_params *P, uint32_t *cg, int *n, int cap)
{
    orc_mz *mq = (orc_mz *)malloc(sizeof(orc_mz) * (size_t)(ql + 8)), *mt = (orc_mz *)malloc(sizeof(orc_mz) * (size_t)(tl + 8));
    int thin, w = seed_window(P->w, ql > tl ? ql : tl, ORC_ALN_SUB_PER, &thin);
    int nq = ql >= P->k ? sketch_thinned(Qb, ql, w, P->k, thin, mq, ql + 8) : 0;
    const char *Ts = (g_ref_seed && Tb >= g_ref && Tb + tl <= g_ref + g_ref_len) ? g_ref_seed + (Tb - g_ref) : Tb;
    int nt = tl >= P->k ? sketch_thinned(Ts, tl, w, P->k, thin, mt, tl + 8) : 0;
    int i = 0, j = 0, na = 0, nch = 0, c;
    anc_t *a; int32_t *f, *pre, *cq, *ct;
    nq = orc_occ_sorted(mq, nq, ORC_ALN_SUB_OCC);
    nt = orc_occ_sorted(mt, nt, ORC_ALN_SUB_OCC);
    a = (anc_t *)malloc(sizeof(anc_t) * ((size_t)(nq < nt ? nq : nt) * ORC_ALN_SUB_OCC + 1));
    while (i < nq && j < nt) {
        if (mq[i].hash < mt[j].hash) i++;
        else if (mq[i].hash > mt[j].hash) j++;
        else {
            int i2 = i, j2 = j, x, y;
            while (i2 < nq && mq[i2].hash == mq[i].hash) i2++;
            while (j2 < nt && mt[j2].hash == mt[j].hash) j2++;
            for (x = i; x < i2; x++) for (y = j; y < j2; y++)
                if (mq[x].rev == mt[y].rev) { a[na].qe = (int32_t)mq[x].pos; a[na].te = (int32_t)mt[y].pos; na++; }
            i = i2; j = j2;
        }
    }
    free(mq); free(mt);
    qsort(a, (size_t)na, sizeof(anc_t), anc_cmp);
    if (na > ORC_ALN_AMAX) na = ORC_ALN_AMAX;       /* the chaining tile of the HIP path */
    f = (int32_t *)malloc(sizeof(int32_t) * ((size_t)na * 2 + 2)); pre = f + na;
    cq = (int32_t *)malloc(sizeof(int32_t) * ((size_t)na + 2) * 2); ct = cq + na + 2;
    cq[0] = -1; ct[0] = -1; nch = 1;                 /* the base pair in front of the box: the outer chain's anchor */
    if (na >= P->min_anchors) {
        int best = chain_dp(a, na, P, f, pre), cnt = 0, k2;
        for (c = best; c >= 0; c = pre[c]) cnt++;
        if (cnt >= P->min_anchors) {
            k2 = 1 + cnt;
            for (c = best; c >= 0; c = pre[c]) { k2--; cq[k2] = a[c].qe; ct[k2] = a[c].te; }
            nch = 1 + cnt;
            while (nch > 1 && (cq[nch - 1] >= ql - 1 || ct[nch - 1] >= tl - 1)) nch--;   /* strictly in front of the last base pair */
        }
    }
    cq[nch] = ql - 1; ct[nch] = tl - 1; nch++;       /* the box's last base pair: the end of the outer chain's next anchor */
    walk(Qb, Tb, cq, ct, nch, 0, ql - 1, tl - 1, P, cg, n, cap, 1);
    free(f); free(a); free(cq);
}

/* the stretch (qbeg .. qend] x (.. tend] along the anchors cq / ct: gap-free runs between anchors stay 'M', the rest is merged
 * into events; query position qbeg and everything up to the first event is 'M' (the caller has extended to qbeg gap-free) */
static void walk(const char *Q, const char *ref, const int32_t *cq, const int32_t *ct, int nch, int qbeg, int qend, int tend,
                 const orc_aln_params *P, uint32_t *cigar, int *n_io, int cigar_cap, int depth)
{
    /* segment classes between consecutive anchors: 0 identical, 1 few mismatches ('M'), 2 needs DP */
    int nseg = nch - 1, s, i, n = *n_io;
    uint8_t *cls = (uint8_t *)malloc((size_t)nseg + 1);
    int mstart_q = qbeg; /* start of the pending M run (query coordinate); the matching ref coordinate follows the diagonal */
    for (s = 0; s < nseg; s++) {
        int dq = cq[s + 1] - cq[s], dt = ct[s + 1] - ct[s], mm = 0, p;
        if (dq != dt) { cls[s] = 2; continue; }
        for (p = 1; p <= dq && mm <= P->max_mm_run; p++) mm += Q[cq[s] + p] != ref[ct[s] + p];
        cls[s] = mm == 0 ? 0 : (mm <= P->max_mm_run ? 1 : 2);
    }
    /* runs of class < 2 are M; runs of class 2 become one padded DP event */
    s = 0;
    while (s < nseg) {
        int e;
        if (cls[s] < 2) { s++; continue; }
        e = s;
        while (e + 1 < nseg && cls[e + 1] == 2) e++;
        {
            /* event covers query (cq[s], cq[e+1]] and ref (ct[s], ct[e+1]] ; pad into the identical neighbours */
            int eqs = cq[s] + 1, eqe = cq[e + 1], ets = ct[s] + 1, ete = ct[e + 1];
            int lp = 0, rp = 0, lim_l, lim_r, tl, ql, nc = 0, sc;
            uint32_t *cg2; uint8_t *bt;
            lim_l = eqs - mstart_q; if (lim_l > P->pad) lim_l = P->pad;
            while (lp < lim_l && Q[eqs - 1 - lp] == ref[ets - 1 - lp]) lp++;
            lim_r = P->pad;
            if (eqe + lim_r > qend) lim_r = qend - eqe;
            if (ete + lim_r > tend) lim_r = tend - ete;
            /* do not pad into the next event: stop at the next class-2 segment's start */
            { int nx = e + 1; while (nx < nseg && cls[nx] < 2) nx++; if (nx < nseg && eqe + lim_r > cq[nx]) lim_r = cq[nx] - eqe; }
            while (rp < lim_r && Q[eqe + 1 + rp] == ref[ete + 1 + rp]) rp++;
            ql = eqe - eqs + 1 + lp + rp; tl = ete - ets + 1 + lp + rp;
            if ((int64_t)ql * tl > P->max_cells) {
                /* too large for the DP: the box between the anchors themselves (no padding) */
                push(cigar, &n, cigar_cap, 0, (uint32_t)(eqs - mstart_q));
                if (depth == 0) sub_align(Q + eqs, eqe - eqs + 1, ref + ets, ete - ets + 1, P, cigar, &n, cigar_cap);
                else corner_event(Q + eqs, eqe - eqs + 1, ref + ets, ete - ets + 1, P, cigar, &n, cigar_cap);
                mstart_q = eqe + 1;
                s = e + 1;
                continue;
            }
            eqs -= lp; ets -= lp; eqe += rp; ete += rp;
            push(cigar, &n, cigar_cap, 0, (uint32_t)(eqs - mstart_q));
            cg2 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(ql + tl + 2));
            bt = (uint8_t *)malloc((size_t)(ql > 0 ? ql : 1) * (size_t)(tl > 0 ? tl : 1));
            if (ql == 0) { nc = 1; cg2[0] = (uint32_t)tl << 4 | 2; }
            else if (tl == 0) { nc = 1; cg2[0] = (uint32_t)ql << 4 | 1; }
            else { sc = orc_nw(ref + ets, tl, Q + eqs, ql, P, cg2, ql + tl + 2, &nc, bt); (void)sc; }
            for (i = 0; i < nc; i++) push(cigar, &n, cigar_cap, cg2[i] & 0xf, cg2[i] >> 4);
            free(cg2); free(bt);
            mstart_q = eqe + 1;
        }
        s = e + 1;
    }
    push(cigar, &n, cigar_cap, 0, (uint32_t)(qend + 1 - mstart_q));
    free(cls);
    *n_io = n;
}

/* one chain -> one record */
static void align_chain(const char *Q, int lenq, const char *ref, int lent, const int32_t *cq, const int32_t *ct, int nch, int rev,
                        const orc_aln_params *P, orc_aln *out, uint32_t *cigar, int cigar_cap)
{
    int i, n = 0;
    int qs0 = cq[0] - P->k + 1, ts0 = ct[0] - P->k + 1; /* first anchor k-mer is part of the alignment */
    int qbeg, tbeg, qend, tend, x, best, bi;
    memset(out, 0, sizeof(*out));
    /* gap-free X-drop extension to the left of the first anchor and to the right of the last one */
    x = 0; best = 0; bi = 0;
    for (i = 1; qs0 - i >= 0 && ts0 - i >= 0; i++) {
        x += pair_score(Q[qs0 - i], ref[ts0 - i], P);
        if (x > best) { best = x; bi = i; }
        if (best - x > P->xdrop) break;
    }
    qbeg = qs0 - bi; tbeg = ts0 - bi;
    x = 0; best = 0; bi = 0;
    for (i = 1; cq[nch - 1] + i < lenq && ct[nch - 1] + i < lent; i++) {
        x += pair_score(Q[cq[nch - 1] + i], ref[ct[nch - 1] + i], P);
        if (x > best) { best = x; bi = i; }
        if (best - x > P->xdrop) break;
    }
    qend = cq[nch - 1] + bi; tend = ct[nch - 1] + bi; /* inclusive */
    push(cigar, &n, cigar_cap, 4, (uint32_t)qbeg);
    walk(Q, ref, cq, ct, nch, qbeg, qend, tend, P, cigar, &n, cigar_cap, 0);
    push(cigar, &n, cigar_cap, 4, (uint32_t)(lenq - 1 - qend));
    shift_gaps_left(cigar, &n, Q, ref, tbeg);
    out->ref_start = tbeg; out->ref_end = tend + 1; out->rev = (uint8_t)rev; out->mapq = 60; out->n_cigar = n;
    out->n_chain = nch; out->q_start = qbeg; out->q_end = qend + 1;
}

/* all records of one contig (the primary first, then the supplementary chains); cigar[r] holds record r's ops from
 * r * cigar_cap.  Returns the number of records. */
int orc_align_contig_multi(const char *contig, int lenq, const char *ref, int lent, const orc_aln_params *P, orc_aln *out,
                           uint32_t *cigar, int cigar_cap, int max_rec)
{
    orc_mz *mq = (orc_mz *)malloc(sizeof(orc_mz) * (size_t)(lenq + 8)), *mt = (orc_mz *)malloc(sizeof(orc_mz) * (size_t)(lent + 8));
    int nq, nt, rev = 0, n_rec, i, r, off = 0, thin, w;
    int chain_n[ORC_ALN_MAX_REC]; uint8_t chain_rev[ORC_ALN_MAX_REC];
    int32_t *cq, *ct;
    char *qrc = NULL, *ref_seed = NULL;
    if (max_rec > ORC_ALN_MAX_REC) max_rec = ORC_ALN_MAX_REC;
    /* long windows: fewer seeds, as fsv_align_batch picks them -- the minimizer window grows with the longer of the two
     * sequences (one seed list of a 50 kb .. 760 kb window stays below the 8 192 the chaining tile holds), and beyond
     * w = 255 the minimizers are thinned by their hash instead (every m-th survives on both sequences alike) */
    w = seed_window(P->w, lenq > lent ? lenq : lent, 3000, &thin);
    nq = sketch_thinned(contig, lenq, w, P->k, thin, mq, lenq + 8);
    {
        char *seed = NULL;
        for (i = 0; i < lent; i++) if (!is_acgt(ref[i])) {
            if (!seed) { seed = (char *)malloc((size_t)lent + 1); memcpy(seed, ref, (size_t)lent); }
            seed[i] = "ACGT"[n_substitute((uint32_t)i)];
        }
        g_ref = ref; g_ref_len = lent; g_ref_seed = seed; ref_seed = seed;
        nt = sketch_thinned(seed ? seed : ref, lent, w, P->k, thin, mt, lent + 8);
    }
    nq = orc_unique_sorted(mq, nq);
    nt = orc_unique_sorted(mt, nt);
    cq = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nq + 1) * 2); ct = cq + nq + 1;
    n_rec = orc_aln_chains(mq, nq, lenq, mt, nt, P, &rev, cq, ct, nq, chain_n, chain_rev, max_rec);
    free(mq); free(mt);
    if (n_rec == 0) { free(cq); free(ref_seed); g_ref_seed = NULL; return 0; }
    for (r = 0; r < n_rec; r++) {
        const char *Q = contig;
        if (chain_rev[r]) {
            if (!qrc) { qrc = (char *)malloc((size_t)lenq); for (i = 0; i < lenq; i++) qrc[i] = comp(contig[lenq - 1 - i]); }
            Q = qrc;
        }
        align_chain(Q, lenq, ref, lent, cq + off, ct + off, chain_n[r], chain_rev[r], P, &out[r], cigar + (size_t)r * cigar_cap, cigar_cap);
        off += chain_n[r];
    }
    free(cq); free(qrc); free(ref_seed); g_ref_seed = NULL;
    return n_rec;
}

int orc_align_contig(const char *contig, int lenq, const char *ref, int lent, const orc_aln_params *P, orc_aln *out,
                     uint32_t *cigar, int cigar_cap)
{
    orc_aln recs[ORC_ALN_MAX_REC];
    uint32_t *cg = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)cigar_cap * ORC_ALN_MAX_REC);
    int n = orc_align_contig_multi(contig, lenq, ref, lent, P, recs, cg, cigar_cap, ORC_ALN_MAX_REC);
    if (n > 0) { *out = recs[0]; memcpy(cigar, cg, sizeof(uint32_t) * (size_t)recs[0].n_cigar); }
    free(cg);
    return n > 0;
}
