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
 * Pipeline: seeds unique in both sequences -> majority strand -> co-linear chain (look-back 64) ->
 * gap-free runs between anchors stay 'M'; everything else is merged into events, padded by up to
 * ALN_PAD identical bases per side (so that a gap can be left-aligned past the seed boundary) and
 * aligned globally with dual-affine gaps; contig ends are extended gap-free with an X-drop and the
 * rest is soft-clipped; last, every gap of the stitched CIGAR is moved to its leftmost position (shift_gaps_left).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static char comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; } }

void orc_aln_default_params(orc_aln_params *P)
{
    P->k = 19; P->w = 19; P->min_anchors = 3; P->lookback = 64; P->max_gap = 20000;
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
                int32_t z = hdiag + (t[i] == q[j - 1] && t[i] != 'N' ? P->a : -P->b);
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

/* chains of (contig, reference); on the reverse strand the contig coordinates are those of its reverse complement.
 * The best chain is the primary alignment.  Like minimap2, which reports what the primary leaves uncovered as supplementary
 * alignments (DipPAV calls SVs beyond the chaining gap from consecutive records of one contig,
 * extract_contig_signature_CCS.py:251-327), the anchors inside the query interval of a chain are then taken out and the rest
 * is chained again, up to ORC_ALN_MAX_REC chains on the contig's majority strand; a supplementary chain needs a score of
 * ORC_ALN_SUP_MIN.  chain_n[r] anchors of chain r follow each other in cq/ct.  Returns the number of chains. */
int orc_aln_chains(const orc_mz *mq, int nq, int lenq, const orc_mz *mt, int nt, const orc_aln_params *P, int *rev_out,
                   int32_t *cq, int32_t *ct, int cap, int *chain_n, int max_rec)
{
    int i = 0, j = 0, n = 0, nf = 0, nr = 0, rev, c, n_rec = 0, used = 0;
    int lim = nq < nt ? nq : nt;
    anc_t *a; uint8_t *sr; int32_t *f, *pre;
    if (lim <= 0) return 0;
    a = (anc_t *)malloc(sizeof(anc_t) * (size_t)lim);
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
    for (i = 0, j = 0; i < n; i++) if (sr[i] == rev) a[j++] = a[i];
    n = j;
    free(sr);
    *rev_out = rev;
    if (n < P->min_anchors) { free(a); return 0; }
    qsort(a, (size_t)n, sizeof(anc_t), anc_cmp);
    f = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * 2); pre = f + n;
    while (n_rec < max_rec && n >= P->min_anchors) {
        int best = -1, cnt = 0, qlo, qhi;
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
        for (c = best; c >= 0; c = pre[c]) cnt++;
        if (cnt < P->min_anchors || used + cnt > cap || (n_rec > 0 && f[best] < ORC_ALN_SUP_MIN)) break;
        { int k2 = used + cnt; for (c = best; c >= 0; c = pre[c]) { k2--; cq[k2] = a[c].qe; ct[k2] = a[c].te; } }
        chain_n[n_rec++] = cnt;
        qlo = cq[used]; qhi = cq[used + cnt - 1];
        used += cnt;
        for (i = 0, j = 0; i < n; i++) if (a[i].qe < qlo || a[i].qe > qhi) a[j++] = a[i]; /* the rest, still in query order */
        n = j;
    }
    free(f); free(a);
    return n_rec;
}

int orc_aln_chain(const orc_mz *mq, int nq, int lenq, const orc_mz *mt, int nt, const orc_aln_params *P, int *rev_out,
                  int32_t *cq, int32_t *ct, int cap)
{
    int cn[1] = {0};
    return orc_aln_chains(mq, nq, lenq, mt, nt, P, rev_out, cq, ct, cap, cn, 1) ? cn[0] : 0;
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
 * Anything that is not A/C/G/T compares as 'A' (the device store has two bits per base). */
static inline char acgt(char c) { return (c == 'C' || c == 'G' || c == 'T') ? c : 'A'; }
int orc_gap_max_shift(const char *s, int off, int len, int cap)
{
    int l = 0;
    while (l < cap && acgt(s[off - 1 - l]) == acgt(s[off + len - 1 - l])) l++;
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
                int l = op == 1 ? orc_gap_max_shift(Q, qoff, (int)len, prev) : orc_gap_max_shift(ref, toff, (int)len, prev);
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

/* one chain -> one record: 1 ok, -1 an event larger than max_cells */
static int align_chain(const char *Q, int lenq, const char *ref, int lent, const int32_t *cq, const int32_t *ct, int nch, int rev,
                       const orc_aln_params *P, orc_aln *out, uint32_t *cigar, int cigar_cap)
{
    int i, n = 0;
    memset(out, 0, sizeof(*out));
    {
        /* segment classes between consecutive anchors: 0 identical, 1 few mismatches ('M'), 2 needs DP */
        int nseg = nch - 1, s;
        uint8_t *cls = (uint8_t *)malloc((size_t)nseg + 1);
        int qs0 = cq[0] - P->k + 1, ts0 = ct[0] - P->k + 1; /* first anchor k-mer is part of the alignment */
        int qbeg, tbeg, qend, tend, x, best, bi;
        for (s = 0; s < nseg; s++) {
            int dq = cq[s + 1] - cq[s], dt = ct[s + 1] - ct[s], mm = 0, p;
            if (dq != dt) { cls[s] = 2; continue; }
            for (p = 1; p <= dq; p++) mm += Q[cq[s] + p] != ref[ct[s] + p];
            cls[s] = mm == 0 ? 0 : (mm <= P->max_mm_run ? 1 : 2);
        }
        /* gap-free X-drop extension to the left of the first anchor and to the right of the last one */
        x = 0; best = 0; bi = 0;
        for (i = 1; qs0 - i >= 0 && ts0 - i >= 0; i++) {
            x += Q[qs0 - i] == ref[ts0 - i] ? P->a : -P->b;
            if (x > best) { best = x; bi = i; }
            if (best - x > P->xdrop) break;
        }
        qbeg = qs0 - bi; tbeg = ts0 - bi;
        x = 0; best = 0; bi = 0;
        for (i = 1; cq[nch - 1] + i < lenq && ct[nch - 1] + i < lent; i++) {
            x += Q[cq[nch - 1] + i] == ref[ct[nch - 1] + i] ? P->a : -P->b;
            if (x > best) { best = x; bi = i; }
            if (best - x > P->xdrop) break;
        }
        qend = cq[nch - 1] + bi; tend = ct[nch - 1] + bi; /* inclusive */
        push(cigar, &n, cigar_cap, 4, (uint32_t)qbeg);
        /* walk the segments: runs of class < 2 are M; runs of class 2 become one padded DP event */
        {
            int mstart_q = qbeg; /* start of the pending M run (query coordinate); the matching ref coordinate follows the diagonal */
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
                    eqs -= lp; ets -= lp; eqe += rp; ete += rp;
                    ql = eqe - eqs + 1; tl = ete - ets + 1;
                    push(cigar, &n, cigar_cap, 0, (uint32_t)(eqs - mstart_q));
                    if ((int64_t)ql * tl > P->max_cells) { free(cls); return -1; }
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
        }
        push(cigar, &n, cigar_cap, 4, (uint32_t)(lenq - 1 - qend));
        shift_gaps_left(cigar, &n, Q, ref, tbeg);
        out->ref_start = tbeg; out->ref_end = tend + 1; out->rev = (uint8_t)rev; out->mapq = 60; out->n_cigar = n;
        out->n_chain = nch; out->q_start = qbeg; out->q_end = qend + 1;
        free(cls);
    }
    return 1;
}

/* all records of one contig (primary first, then supplementary chains); cigar[r] holds record r's ops from r * cigar_cap.
 * returns the number of records, -1 when an event exceeds max_cells */
int orc_align_contig_multi(const char *contig, int lenq, const char *ref, int lent, const orc_aln_params *P, orc_aln *out,
                           uint32_t *cigar, int cigar_cap, int max_rec)
{
    orc_mz *mq = (orc_mz *)malloc(sizeof(orc_mz) * (size_t)(lenq + 8)), *mt = (orc_mz *)malloc(sizeof(orc_mz) * (size_t)(lent + 8));
    int nq, nt, rev = 0, n_rec, i, r, w = P->w, off = 0, rc = 0;
    int chain_n[ORC_ALN_MAX_REC];
    int32_t *cq, *ct;
    char *q = NULL;
    const char *Q;
    if (max_rec > ORC_ALN_MAX_REC) max_rec = ORC_ALN_MAX_REC;
    {
        /* long windows: fewer seeds, as fsv_align_batch picks them -- the minimizer window grows with the longer of the two
         * sequences (one seed list of a 50 kb .. 760 kb window stays below the 8 192 the chaining tile holds), and beyond
         * w = 255 the minimizers are thinned by their hash instead (every m-th survives on both sequences alike) */
        const int L = lenq > lent ? lenq : lent;
        int m = 1;
        if (L / 3000 + 1 > w) w = L / 3000 + 1;
        if (w > 255) { m = (w + 254) / 255; w = 255; }
        nq = orc_sketch(contig, lenq, w, P->k, 0, mq, lenq + 8);
        nt = orc_sketch(ref, lent, w, P->k, 0, mt, lent + 8);
        if (m > 1) {
            int j = 0;
            for (i = 0; i < nq; i++) if ((mq[i].hash >> 11) % (uint64_t)m == 0) mq[j++] = mq[i];
            nq = j; j = 0;
            for (i = 0; i < nt; i++) if ((mt[i].hash >> 11) % (uint64_t)m == 0) mt[j++] = mt[i];
            nt = j;
        }
        nq = orc_unique_sorted(mq, nq);
        nt = orc_unique_sorted(mt, nt);
    }
    cq = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nq + 1) * 2); ct = cq + nq + 1;
    n_rec = orc_aln_chains(mq, nq, lenq, mt, nt, P, &rev, cq, ct, nq, chain_n, max_rec);
    free(mq); free(mt);
    if (n_rec == 0) { free(cq); return 0; }
    if (rev) { q = (char *)malloc((size_t)lenq); for (i = 0; i < lenq; i++) q[i] = comp(contig[lenq - 1 - i]); Q = q; } else Q = contig;
    for (r = 0; r < n_rec && rc >= 0; r++) {
        rc = align_chain(Q, lenq, ref, lent, cq + off, ct + off, chain_n[r], rev, P, &out[r], cigar + (size_t)r * cigar_cap, cigar_cap);
        off += chain_n[r];
    }
    free(cq); free(q);
    return rc < 0 ? -1 : n_rec;
}

int orc_align_contig(const char *contig, int lenq, const char *ref, int lent, const orc_aln_params *P, orc_aln *out,
                     uint32_t *cigar, int cigar_cap)
{
    int n = orc_align_contig_multi(contig, lenq, ref, lent, P, out, cigar, cigar_cap, 1);
    return n < 0 ? -1 : (n > 0);
}
