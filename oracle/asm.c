/* oracle/asm.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, ASCII reads) of the per-read-set local
 * assembly that FocalSV obtains from `hifiasm -o X.asm -t T X.fa`
 * (focalsv/3_assembly/run_assembly.py:15-26), stage by stage:
 *
 *   S1 sketch            ha_sketch, k=w=51 HPC            sketch.cpp:39-137          (oracle/sketch.c)
 *   S2 per-read index    occurrence filter                htab.cpp:917-998 (ha_ft_gen/ha_pt_gen) -- restated as
 *                        "hash occurs once in the read"; the global count filter needs a whole-genome histogram
 *   S3 anchors + chain   ha_get_candidates_interface      anchor.cpp:60-178, 207-300
 *                        chain_DP                         Hash_Table.cpp:425-616
 *                        extension to the read ends       Hash_Table.cpp:83-243
 *   S4 window verify     verify_window + K5               Correct.cpp:203-250, 306-531     (oracle/bpm.c)
 *   S5 rescue + accept   recalcate_window_advance         Correct.cpp:2629-3023 (right-extension pass, 0.9 filter,
 *                        non_trim_error_rate <= 0.03      Correct.cpp:725)
 *   S6 window paths      Reserve_Banded_BPM_PATH + generate_cigar   (oracle/bpm.c)
 *   S7 consensus         generate_consensus/window_consensus/get_seq_from_Graph  Correct.cpp:4731-4808, 4132-4195, 4010-4130
 *                        restated as a per-column vote: thresholds 0.60 / 0.515 in homopolymers,
 *                        coverage >= 3 (Correct.h:11-15)
 *   S8 apply + RC        worker_ec_save                   Assembly.cpp:706-767 (reads are reverse-complemented after
 *                        every non-final round, :747-757)
 *   S9 final overlaps    worker_ov_final / if_exact_match Assembly.cpp:1284-1306, 894-974
 *   S10 layout           containment removal, best-overlap string graph, unitig walk;
 *                        contig = concatenated read prefixes (ma_ug_seq, Overlaps.cpp:8969-9034)
 *
 * All decisions are integer arithmetic so that the HIP path can be compared bit for bit.
 * Deviations from hifiasm that are deliberate and documented in DESIGN.md: no global k-mer
 * count table (S2), a fixed chain look-back instead of max_skip heuristics, no left-extension
 * rescue / fix_boundary / boundary re-alignment (junction insertions are voted directly),
 * no haplotype partition (K7) and no bubble/tip cleaning (read sets are single-haplotype).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "oracle.h"

/* ---------------------------------------------------------------- helpers */
static char comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; } }

static void revcomp_inplace(char *s, int n)
{
    int i;
    for (i = 0; i < n / 2; i++) { char a = comp(s[i]), b = comp(s[n - 1 - i]); s[i] = b; s[n - 1 - i] = a; }
    if (n & 1) s[n / 2] = comp(s[n / 2]);
}

/* base of read y at strand coordinate p */
static inline char ybase(const char *y, int ylen, int rev, int p) { return rev ? comp(y[ylen - 1 - p]) : y[p]; }

int orc_thr_for_len_p(const orc_asm_params *P, int x_len)
{
    /* verify_window: threshold = x_len * max_ov_diff_ec (0.04), Adjust_Threshold (Correct.h:39) */
    int t;
    const double rate = P->win_rate_pm / 1000.0;      /* 40 / 1000.0 is the double 0.04 */
    if (x_len == ORC_WINDOW) return P->win_rate_pm == 40 ? ORC_K_FULL : (int)(ORC_WINDOW * rate);
    t = (int)(x_len * rate);
    if (t == 0 && x_len >= 4) t = 1;
    return t;
}

int orc_double_thr_p(const orc_asm_params *P, int pre, int x_len)
{
    /* double_error_threshold, Correct.cpp:658-676 */
    int t;
    if (pre == 0 && x_len >= 4) pre = 1;
    t = pre * 2;
    if (x_len >= 300 && t < P->k_cap) t = P->k_cap;
    if (t > P->k_cap) t = P->k_cap;
    return t;
}

static const orc_asm_params *hifi_params(void)
{
    static orc_asm_params P; static int init = 0;
    if (!init) { orc_asm_default_params(&P); init = 1; }
    return &P;
}
int orc_thr_for_len(int x_len) { return orc_thr_for_len_p(hifi_params(), x_len); }
int orc_double_thr(int pre, int x_len) { return orc_double_thr_p(hifi_params(), pre, x_len); }

/* ---------------------------------------------------------------- S2: unique-in-read minimizers sorted by hash */
static int mz_cmp(const void *a, const void *b)
{
    const orc_mz *x = (const orc_mz *)a, *y = (const orc_mz *)b;
    if (x->hash != y->hash) return x->hash < y->hash ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return 0;
}

int orc_unique_sorted(orc_mz *mz, int n)
{
    int i, j, m = 0;
    qsort(mz, (size_t)n, sizeof(orc_mz), mz_cmp);
    for (i = 0; i < n; i = j) {
        for (j = i + 1; j < n && mz[j].hash == mz[i].hash; j++) {}
        if (j - i == 1) mz[m++] = mz[i];
    }
    return m;
}

/* ---------------------------------------------------------------- S3: anchors + chain for one ordered pair */
typedef struct { int32_t qe, te; } anchor_t;

static int anchor_cmp(const void *a, const void *b)
{
    const anchor_t *x = (const anchor_t *)a, *y = (const anchor_t *)b;
    if (x->qe != y->qe) return x->qe < y->qe ? -1 : 1;
    if (x->te != y->te) return x->te < y->te ? -1 : 1;
    return 0;
}

/* returns 1 and fills *o (+ chain anchors, sorted by qe) when the pair overlaps */
int orc_chain_pair(const orc_mz *q, int nq, int lenq, const orc_mz *t, int nt, int lent, const orc_asm_params *P,
                   int bw_per_mille, orc_ovl *o, int32_t *chain_qe, int32_t *chain_te, int chain_cap)
{
    anchor_t *a;
    int32_t *f, *pre, *ind, *sl;
    int i, j, n = 0, nf = 0, nr = 0, rev, best = -1, ok = 0;
    int cap = nq < nt ? nq : nt;
    if (cap <= 0) return 0;
    a = (anchor_t *)malloc(sizeof(anchor_t) * (size_t)cap * 2);
    {
        /* merge join on hash (both lists sorted, hashes unique inside each read) */
        uint8_t *srev = (uint8_t *)malloc((size_t)cap);
        anchor_t *raw = a + cap;
        int32_t *tspan = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
        int32_t *qspan = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
        i = j = 0;
        while (i < nq && j < nt) {
            if (q[i].hash < t[j].hash) i++;
            else if (q[i].hash > t[j].hash) j++;
            else {
                srev[n] = q[i].rev ^ t[j].rev;
                raw[n].qe = (int32_t)q[i].pos;
                raw[n].te = (int32_t)t[j].pos;
                tspan[n] = t[j].span;
                qspan[n] = q[i].span;
                if (srev[n]) nr++; else nf++;
                n++; i++; j++;
            }
        }
        rev = nr > nf;
        for (i = 0, j = 0; i < n; i++) {
            if (srev[i] != rev) continue;
            a[j].qe = raw[i].qe;
            a[j].te = rev ? (lent - 1) - (raw[i].te - tspan[i] + 1) : raw[i].te;
            if (rev) {   /* as hifiasm chains a reverse-strand pair: the query on its reverse strand (k-mer ends there), the target
                          * forward (Hash_Table.cpp:619-676, x_pos_strand = 1) -- the chain's indel budget runs from that end */
                a[j].qe = (lenq - 1) - (raw[i].qe - qspan[i] + 1);
                a[j].te = raw[i].te;
            }
            j++;
        }
        n = j;
        free(srev); free(tspan); free(qspan);
    }
    if (n < P->min_anchors) { free(a); return 0; }
    qsort(a, (size_t)n, sizeof(anchor_t), anchor_cmp);

    f = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * 4);
    pre = f + n; ind = pre + n; sl = ind + n;
    for (i = 0; i < n; i++) {
        int32_t bs = P->k, bp = -1, bi = 0, bl = 0;
        int lo = i - P->lookback < 0 ? 0 : i - P->lookback;
        for (j = i - 1; j >= lo; j--) { /* nearest predecessor first; strict '>' keeps the nearest on ties */
            int32_t dq = a[i].qe - a[j].qe, dt = a[i].te - a[j].te, gap, ti, tl, sc;
            if (dq <= 0 || dt <= 0) continue;
            gap = dq > dt ? dq - dt : dt - dq;
            ti = ind[j] + gap; tl = sl[j] + dq;
            if ((int64_t)ti * 1000 > (int64_t)tl * bw_per_mille) continue; /* bw 0 = co-linear anchors only */
            sc = dq < dt ? dq : dt;
            if (sc > P->k) sc = P->k;
            if (ti) sc -= (int32_t)(((int64_t)ti * sc * 1000) / ((int64_t)tl * bw_per_mille));
            sc += f[j];
            if (sc > bs) { bs = sc; bp = j; bi = ti; bl = tl; }
        }
        f[i] = bs; pre[i] = bp; ind[i] = bi; sl[i] = bl;
    }
    for (i = 0; i < n; i++) if (best < 0 || f[i] > f[best]) best = i;
    {
        int cnt = 0, c;
        for (c = best; c >= 0; c = pre[c]) cnt++;
        if (cnt >= P->min_anchors && cnt <= chain_cap) {
            int k2 = cnt;
            for (c = best; c >= 0; c = pre[c]) { k2--; chain_qe[k2] = a[c].qe; chain_te[k2] = a[c].te; }
            {
                int32_t xs = chain_qe[0], ys = chain_te[0], xe = chain_qe[cnt - 1], ye = chain_te[cnt - 1], m, r;
                m = xs < ys ? xs : ys; xs -= m; ys -= m;
                r = (lenq - 1 - xe) < (lent - 1 - ye) ? (lenq - 1 - xe) : (lent - 1 - ye);
                xe += r; ye += r;
                if (rev) {   /* back to the query forward / target on its reverse strand; anchors in query order */
                    int32_t t0;
                    t0 = xs; xs = (lenq - 1) - xe; xe = (lenq - 1) - t0;
                    t0 = ys; ys = (lent - 1) - ye; ye = (lent - 1) - t0;
                    for (c = 0; c < cnt; c++) { chain_qe[c] = (lenq - 1) - chain_qe[c]; chain_te[c] = (lent - 1) - chain_te[c]; }
                    for (c = 0; c < cnt / 2; c++) {
                        t0 = chain_qe[c]; chain_qe[c] = chain_qe[cnt - 1 - c]; chain_qe[cnt - 1 - c] = t0;
                        t0 = chain_te[c]; chain_te[c] = chain_te[cnt - 1 - c]; chain_te[cnt - 1 - c] = t0;
                    }
                }
                if (xe - xs + 1 >= P->min_ovlp) {
                    memset(o, 0, sizeof(*o));
                    o->x_s = xs; o->x_e = xe; o->y_s = ys; o->y_e = ye; o->rev = (uint8_t)rev;
                    o->score = f[best]; o->n_chain = cnt;
                    ok = 1;
                }
            }
        }
    }
    free(f); free(a);
    return ok;
}

/* diagonal (te - qe) predicted for query position x: last chain anchor with qe <= x, else the first anchor */
static int32_t diag_at(const int32_t *cq, const int32_t *ct, int n, int32_t x)
{
    int lo = 0, hi = n; /* first index with cq > x */
    while (lo < hi) { int mid = (lo + hi) >> 1; if (cq[mid] <= x) lo = mid + 1; else hi = mid; }
    if (lo == 0) return ct[0] - cq[0];
    return ct[lo - 1] - cq[lo - 1];
}

/* ---------------------------------------------------------------- S4: one window */
/* determine_overlap_region + fill_subregion + K5.  Returns 1 when the window is geometrically valid. */
static int window_verify(const char *x, const char *y, int ylen, int rev, orc_win *w, char *ybuf, int k_cap)
{
    int n = w->x_len, k = w->k, wlen = n + 2 * k, j, win0, err;
    w->end_site = -1; w->err = -1; w->y_beg = -1; w->extra_begin = -1; w->extra_end = -1;
    if (w->y_start < 0 || ylen <= w->y_start || ylen - w->y_start + 2 * k + k_cap < wlen) return 0;
    win0 = w->y_start - k;
    {
        int ys = win0, olen = wlen < ylen - ys ? wlen : ylen - ys;
        w->extra_end = (int16_t)(wlen - olen); w->extra_begin = 0;
        if (ys < 0) { w->extra_begin = (int16_t)(-ys); ys = 0; }
        w->y_beg = ys;
    }
    for (j = 0; j < wlen; j++) { int p = win0 + j; ybuf[j] = (p < 0 || p >= ylen) ? 'N' : ybase(y, ylen, rev, p); }
    w->end_site = k <= ORC_K_MAX ? orc_bpm(ybuf, wlen, x + w->x_start, n, k, &err) : orc_bpm_wide(ybuf, wlen, x + w->x_start, n, k, &err);
    w->err = err;
    if (err < 0) w->end_site = -1;
    return 1;
}

/* absolute strand coordinate of padded column c */
static inline int win_abs(const orc_win *w, int c) { return w->y_start - w->k + c; }

/* non_trim_error_rate's charge for the unmatched window j of an overlap (Correct.cpp:745-838): the alignment of the window before it is
 * extended into it from the left, that of the window after it from the right (verify_sub_window :675-722: the doubled threshold,
 * Reserve_Banded_BPM_Extension, for the right side on the reversed strings); charged are the two distances and the bases neither reaches.
 * Returns the running total (the reference adds to a long long, through float arithmetic in one branch). */
static long unmatched_charge(const char *x, const char *y, int ylen, int rev, const orc_win *W, int n_win, int j, const orc_asm_params *P,
                             char *ybuf, long terr)
{
    const orc_win *w = &W[j];
    const int n = w->x_len, thr = orc_double_thr_p(P, w->k, n), wlen = n + 2 * thr;
    int yb[2] = {-1, -1}, al[2] = {0, 0}, er[2] = {0, 0}, d, c;
    static __thread char xr[ORC_WINDOW + 8], yr[ORC_WINDOW + 2 * ORC_K_WIDE + 8];
    if (w->y_beg < 0) return terr + n;                               /* (w_list[i].y_start == -1: the window lies outside y) */
    if (thr > ORC_K_MAX) return terr + n;                            /* (wide-band profiles: not restated) */
    if (j > 0 && W[j - 1].err >= 0) yb[0] = W[j - 1].ry_end + 1;
    if (j + 1 < n_win && W[j + 1].err >= 0) yb[1] = W[j + 1].ry_start - n;
    if (yb[0] < 0 && yb[1] < 0) yb[0] = yb[1] = w->y_beg + w->k - w->extra_begin;
    if (yb[0] < 0) yb[0] = yb[1];
    if (yb[1] < 0) yb[1] = yb[0];
    for (d = 0; d < 2; d++) {
        const int win0 = yb[d] - thr;
        int e, pe, te;
        if (yb[d] < 0 || ylen <= yb[d] || ylen - yb[d] + 2 * thr + P->k_cap < wlen) continue;    /* determine_overlap_region */
        for (c = 0; c < wlen; c++) { const int p = win0 + c; ybuf[c] = (p < 0 || p >= ylen) ? 'N' : ybase(y, ylen, rev, p); }
        if (d == 0) te = orc_bpm_extension(ybuf, x + w->x_start, n, thr, &e, &pe);
        else {
            for (c = 0; c < wlen; c++) yr[c] = ybuf[wlen - 1 - c];
            for (c = 0; c < n; c++) xr[c] = x[w->x_start + n - 1 - c];
            te = orc_bpm_extension(yr, xr, n, thr, &e, &pe);
        }
        if (te >= 0) { al[d] = te + 1; er[d] = e; }
    }
    if (al[0] && al[1]) {
        if (al[0] + al[1] <= n) return terr + er[0] + er[1] + (n - al[0] - al[1]);
        { const float rate = (float)n / (float)(al[0] + al[1]); return (long)((float)terr + (float)(unsigned)(er[0] + er[1]) * rate); }
    }
    if (!al[0] && !al[1]) return terr + n;
    return al[0] ? terr + er[0] + (n - al[0]) : terr + er[1] + (n - al[1]);
}

/* fix_boundary applies to the windows' final cigars and to the left-extension pass, not to the junction alignments */
static __thread int g_fix_boundary = 0;

/* ---------------------------------------------------------------- S6: path of one matched window */
/* Fills w->path (start-to-end ops), w->path_len, w->ry_start / w->ry_end (absolute, inclusive), w->err (after gap shifting). */
static void window_path(const char *x, const char *y, int ylen, int rev, orc_win *w, char *ybuf, uint64_t *cols, uint8_t *tmp,
                        int *rl, uint8_t *ro)
{
    int n = w->x_len, k = w->k, wlen = n + 2 * k, j, win0 = w->y_start - k, err, start = 0, plen = 0, end, nrun, i, pl;
    for (j = 0; j < wlen; j++) { int p = win0 + j; ybuf[j] = (p < 0 || p >= ylen) ? 'N' : ybase(y, ylen, rev, p); }
    end = w->end_site; err = w->err;
    if (err == 0) {
        start = end - n + 1;
        for (i = 0; i < n; i++) tmp[i] = 0;
        plen = n;
    } else if (!orc_try_cigar(ybuf, x + w->x_start, n, end, err, tmp, &start, &plen)) {
        int e2;
        end = k <= ORC_K_MAX ? orc_bpm_path(ybuf, wlen, x + w->x_start, n, k, &e2, &start, &plen, tmp, cols)
                             : orc_bpm_path_wide(ybuf, wlen, x + w->x_start, n, k, &e2, &start, &plen, tmp, cols);
        err = e2;
    }
    /* fix_boundary (Correct.cpp:1676-1795; called for the windows' final cigars :2968 and in the left-extension pass :2858): an alignment
     * that starts in the first column of its padded window, or ends in its last one, may have been cut off by the band -- the window
     * is aligned once more with the band shifted by k towards that side (from the old region's first base / so that the x interval
     * ends at the old alignment's last base), without a hint, and the new alignment stands when it has fewer errors */
    if (g_fix_boundary && err > 0 && k <= ORC_K_MAX && (start == 0 || end == wlen - 1)) {
        orc_win t2 = *w;
        int ok = 0;
        if (start == 0) { if (w->extra_begin == 0) { t2.y_start = w->y_beg; ok = 1; } }
        else if (w->extra_end == 0) { t2.y_start = (w->y_beg + end) - n + 1; ok = 1; }     /* (total_y_start + local_y_end: extra_begin is not taken off, as in the reference) */
        if (ok && t2.y_start >= 0 && ylen > t2.y_start && !(ylen - t2.y_start + 2 * k + ORC_K_MAX < wlen)) {
            const int win2 = t2.y_start - k, yb2 = win2 < 0 ? 0 : win2;
            if (yb2 != w->y_beg) {
                char ybuf2[ORC_WINDOW + 2 * ORC_K_MAX + 8];
                static __thread uint8_t tmp2[2 * ORC_WINDOW + 4 * ORC_K_WIDE + 16];
                int e2, start2 = 0, plen2 = 0, end2, olen2 = wlen < ylen - win2 ? wlen : ylen - win2;
                for (j = 0; j < wlen; j++) { int p = win2 + j; ybuf2[j] = (p < 0 || p >= ylen) ? 'N' : ybase(y, ylen, rev, p); }
                end2 = orc_bpm_path(ybuf2, wlen, x + w->x_start, n, k, &e2, &start2, &plen2, tmp2, cols);
                if (getenv("ORC_DEBUG_FIX")) fprintf(stderr, "FIX x_start %d n %d k %d start %d end %d err %d -> err %d%s\n", w->x_start, n, k, start, end, err, e2, e2 >= 0 && e2 < err ? " adopted" : "");
                if (e2 >= 0 && e2 < err) {
                    w->y_start = t2.y_start; w->y_beg = yb2; w->extra_begin = (int16_t)(win2 < 0 ? -win2 : 0); w->extra_end = (int16_t)(wlen - olen2);
                    w->end_site = end2; end = end2; err = e2; start = start2; plen = plen2;
                    memcpy(tmp, tmp2, (size_t)plen2); memcpy(ybuf, ybuf2, (size_t)wlen);
                }
            }
        }
    }
    if (err > 0) nrun = orc_generate_cigar(tmp, plen, n, x + w->x_start, ybuf, &start, &end, &err, rl, ro);
    else { nrun = 1; rl[0] = n; ro[0] = 0; }
    pl = 0;
    for (i = 0; i < nrun; i++) pl += rl[i];
    if (pl > ORC_PATH_REC) {   /* the 128-byte path record of the HIP path holds 416 ops (x_len + k <= 406 for hifiasm's thresholds: never
                                  reached); a longer path -- a wide-band window with more than 41 inserted bases -- leaves the window unused */
        w->path_len = 0; w->err = -1;
        return;
    }
    pl = 0;
    for (i = 0; i < nrun; i++) for (j = 0; j < rl[i]; j++) w->path[pl++] = ro[i];
    w->path_len = (int16_t)pl;
    w->err = err;
    w->ry_start = win_abs(w, start);
    w->ry_end = win_abs(w, end);
}

/* ---------------------------------------------------------------- S7: consensus of one grid window of read x */
/* insertion votes are kept as an event list (column, key); key = len << 24 | 2-bit bases, len <= INS_MAXLEN */
typedef struct { int32_t col; uint32_t key; } ins_event;
#define INS_MAXLEN 12

static inline int base2(char c) { switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 0; } }

typedef struct { ins_event *e; int n, cap; } ins_list;
static void ins_vote(ins_list *l, int col, uint32_t key)
{
    if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 256; l->e = (ins_event *)realloc(l->e, sizeof(ins_event) * (size_t)l->cap); }
    l->e[l->n].col = col; l->e[l->n].key = key; l->n++;
}
/* most frequent key at a column; ties go to the smaller key; returns its count (0 = none) */
static int ins_winner(const ins_list *l, int col, uint32_t *key_out)
{
    int i, j, best = 0; uint32_t bk = 0;
    for (i = 0; i < l->n; i++) {
        int c = 0;
        if (l->e[i].col != col) continue;
        for (j = 0; j < l->n; j++) c += (l->e[j].col == col && l->e[j].key == l->e[i].key);
        if (c > best || (c == best && l->e[i].key < bk)) { best = c; bk = l->e[i].key; }
    }
    *key_out = bk;
    return best;
}

static int is_homopolymer_site(const char *x, int xlen, int p)
{
    /* if_is_homopolymer_strict (Correct.h:447-530), statement by statement: the run that starts right after the site and the run that
     * starts right before it, each looked at over at most three bases; the site joins the forward run if it has that base, else the
     * backward run if it has that one; a run of three (the site included or merely beside it) makes the site a homopolymer site,
     * and so do a forward and a backward run of the site's own base that add up to three */
    const int threshold = 3;
    int beg = p - threshold < 0 ? 0 : p - threshold, end = p + threshold >= xlen ? xlen - 1 : p + threshold, i;
    char f_ch = 0, b_ch = 0;
    int f_len = 0, b_len = 0;
    for (i = p + 1; i <= end; i++) {
        if (f_ch == 0) { f_ch = x[i]; f_len = 1; }
        else if (x[i] != f_ch) break;
        else f_len++;
    }
    for (i = p - 1; i >= beg; i--) {
        if (b_ch == 0) { b_ch = x[i]; b_len = 1; }
        else if (x[i] != b_ch) break;
        else b_len++;
    }
    if (f_ch == x[p]) f_len++;
    else if (b_ch == x[p]) b_len++;
    if (f_len >= threshold || b_len >= threshold) return 1;
    if (x[p] == f_ch && b_ch == f_ch && f_len + b_len >= threshold) return 1;
    return 0;
}

static inline int wins(int cnt, int total, int homo)
{
    if (cnt * 5 >= total * 3) return 1;                  /* >= 0.60 */
    if (homo && cnt * 1000 >= total * 515) return 1;     /* >= 0.515 inside a homopolymer */
    return 0;
}

/* ---------------------------------------------------------------- one correction round over a read set */
typedef struct {
    int n;
    char **seq;
    int *len;
} readset;

static int ovl_cmp(const void *a, const void *b)
{
    const orc_ovl *x = (const orc_ovl *)a, *y = (const orc_ovl *)b;
    if (x->q != y->q) return x->q < y->q ? -1 : 1;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return 0;
}

/* An overlap is a symmetric object: every unordered pair (a < b) is chained once, with a as the query, and the overlap of
 * b on a is the mirror image of the same chain (hifiasm recomputes it from b's side, anchor.cpp:207-300; mirroring halves the
 * chaining work and both directions agree on the anchors).  On the reverse strand the mirrored coordinates are measured from
 * the other end of both reads and the anchor order flips. */
/* both_ways (the final pass's gapped re-chain): the overlap of b on a is chained from b's side as hifiasm does -- with an indel
 * budget the chain DP depends on the end it starts from (the budget is a rate over the span chained so far, and on the reverse
 * strand the two sides start from opposite ends), so the mirror image can differ by the bases of an indel near a read end */
static __thread int g_both_ways = 0;
static void collect_overlaps(const readset *R, const orc_asm_params *P, int bw, orc_mz **uq, int *nuq, orc_ovl **ovl_out,
                             int32_t **cq_out, int32_t **ct_out, int *n_out)
{
    int q, t, n = 0, cap = 1024, chain_cap = 0, ccap = 1 << 16, cused = 0, i;
    orc_ovl *ov = (orc_ovl *)malloc(sizeof(orc_ovl) * (size_t)cap);
    int32_t *cq = (int32_t *)malloc(sizeof(int32_t) * (size_t)ccap), *ct = (int32_t *)malloc(sizeof(int32_t) * (size_t)ccap);
    for (q = 0; q < R->n; q++) if (nuq[q] > chain_cap) chain_cap = nuq[q];
    for (q = 0; q < R->n; q++)
        for (t = q + 1; t < R->n; t++) {
            orc_ovl o, m;
            const int lenq = R->len[q], lent = R->len[t];
            if (cused + 2 * chain_cap > ccap) {
                ccap = (cused + 2 * chain_cap) * 2;
                cq = (int32_t *)realloc(cq, sizeof(int32_t) * (size_t)ccap);
                ct = (int32_t *)realloc(ct, sizeof(int32_t) * (size_t)ccap);
            }
            if (!orc_chain_pair(uq[q], nuq[q], lenq, uq[t], nuq[t], lent, P, bw, &o, cq + cused, ct + cused, chain_cap)) continue;
            o.q = (uint32_t)q; o.t = (uint32_t)t; o.chain_off = cused;
            m = o;
            m.q = (uint32_t)t; m.t = (uint32_t)q; m.chain_off = cused + o.n_chain;
            if (!o.rev) {
                m.x_s = o.y_s; m.x_e = o.y_e; m.y_s = o.x_s; m.y_e = o.x_e;
                for (i = 0; i < o.n_chain; i++) { cq[m.chain_off + i] = ct[o.chain_off + i]; ct[m.chain_off + i] = cq[o.chain_off + i]; }
            } else {
                m.x_s = lent - 1 - o.y_e; m.x_e = lent - 1 - o.y_s; m.y_s = lenq - 1 - o.x_e; m.y_e = lenq - 1 - o.x_s;
                for (i = 0; i < o.n_chain; i++) {
                    const int src = o.chain_off + o.n_chain - 1 - i;
                    cq[m.chain_off + i] = lent - 1 - ct[src]; ct[m.chain_off + i] = lenq - 1 - cq[src];
                }
            }
            if (g_both_ways) {
                orc_ovl m2;
                if (n + 2 > cap) { cap *= 2; ov = (orc_ovl *)realloc(ov, sizeof(orc_ovl) * (size_t)cap); }
                ov[n++] = o;
                cused += o.n_chain;
                if (getenv("ORC_DEBUG_CHAIN")) {
                    int dq = -1, dt = -1, z;
                    sscanf(getenv("ORC_DEBUG_CHAIN"), "%d,%d", &dq, &dt);
                    if (dq == q && dt == t) { fprintf(stderr, "CHAIN %d->%d rev %d x [%d,%d] y [%d,%d] n %d:", q, t, o.rev, o.x_s, o.x_e, o.y_s, o.y_e, o.n_chain); for (z = 0; z < o.n_chain; z++) fprintf(stderr, " (%d,%d)", cq[o.chain_off + z], ct[o.chain_off + z]); fprintf(stderr, "\n"); }
                }
                if (orc_chain_pair(uq[t], nuq[t], lent, uq[q], nuq[q], lenq, P, bw, &m2, cq + cused, ct + cused, chain_cap)) {
                    if (getenv("ORC_DEBUG_CHAIN")) {
                        int dq = -1, dt = -1, z;
                        sscanf(getenv("ORC_DEBUG_CHAIN"), "%d,%d", &dq, &dt);
                        if (dq == t && dt == q) { fprintf(stderr, "CHAIN %d->%d rev %d x [%d,%d] y [%d,%d] n %d:", t, q, m2.rev, m2.x_s, m2.x_e, m2.y_s, m2.y_e, m2.n_chain); for (z = 0; z < m2.n_chain; z++) fprintf(stderr, " (%d,%d)", cq[cused + z], ct[cused + z]); fprintf(stderr, "\n"); }
                    }
                    m2.q = (uint32_t)t; m2.t = (uint32_t)q; m2.chain_off = cused;
                    cused += m2.n_chain;
                    ov[n++] = m2;
                }
                continue;
            }
            cused += 2 * o.n_chain;
            if (n + 2 > cap) { cap *= 2; ov = (orc_ovl *)realloc(ov, sizeof(orc_ovl) * (size_t)cap); }
            ov[n++] = o; ov[n++] = m;
        }
    qsort(ov, (size_t)n, sizeof(orc_ovl), ovl_cmp); /* the consumers expect the overlaps of one query to be contiguous */
    *ovl_out = ov; *cq_out = cq; *ct_out = ct; *n_out = n;
}

static void sketch_set(const readset *R, const orc_asm_params *P, int w, orc_mz ***uq_out, int **nuq_out)
{
    int r;
    orc_mz **uq = (orc_mz **)malloc(sizeof(orc_mz *) * (size_t)R->n);
    int *nuq = (int *)malloc(sizeof(int) * (size_t)R->n);
    for (r = 0; r < R->n; r++) {
        int cap = R->len[r] + 8, n;
        uq[r] = (orc_mz *)malloc(sizeof(orc_mz) * (size_t)cap);
        n = orc_sketch(R->seq[r], R->len[r], w, P->k, P->hpc, uq[r], cap);
        nuq[r] = orc_unique_sorted(uq[r], n);
    }
    *uq_out = uq; *nuq_out = nuq;
}

static void free_sketch(const readset *R, orc_mz **uq, int *nuq)
{
    int r;
    for (r = 0; r < R->n; r++) free(uq[r]);
    free(uq); free(nuq);
}

/* Build + verify + rescue + accept + path every window of every overlap of the set.
 * Returns the window array (owned by caller); ov[].first_win/n_win index into it. */
static orc_win *align_overlaps(const readset *R, const orc_asm_params *P, orc_ovl *ov, int n_ov, const int32_t *cq, const int32_t *ct, int *n_win_out)
{
    const int fix_saved_ = (g_fix_boundary = P->fix_boundary, 0);
    int i, j, total = 0, wi;
    orc_win *W;
    char ybuf[ORC_WINDOW + 2 * ORC_K_WIDE + 8];
    static __thread uint64_t cols[5 * 4 * (ORC_WINDOW + 4)];
    uint8_t tmp[2 * ORC_WINDOW + 4 * ORC_K_WIDE + 16];
    int rl[2 * ORC_WINDOW + 64];
    uint8_t ro[2 * ORC_WINDOW + 64];
    for (i = 0; i < n_ov; i++) {
        ov[i].first_win = total;
        ov[i].n_win = ov[i].x_e / ORC_WINDOW - ov[i].x_s / ORC_WINDOW + 1;
        total += ov[i].n_win;
    }
    W = (orc_win *)calloc((size_t)total + 1, sizeof(orc_win));
    for (i = 0; i < n_ov; i++) {
        orc_ovl *o = &ov[i];
        const char *x = R->seq[o->q], *y = R->seq[o->t];
        int ylen = R->len[o->t], w0 = o->x_s / ORC_WINDOW;
        int64_t tlen = 0, terr = 0;
        o->align_len = 0;
        for (j = 0; j < o->n_win; j++) {
            orc_win *w = &W[o->first_win + j];
            int gs = (w0 + j) * ORC_WINDOW, ge = gs + ORC_WINDOW - 1;
            w->ovl = (uint32_t)i; w->win = (uint32_t)j;
            w->x_start = gs > o->x_s ? gs : o->x_s;
            w->x_len = (int16_t)((ge < o->x_e ? ge : o->x_e) - w->x_start + 1);
            w->k = (uint8_t)orc_thr_for_len_p(P, w->x_len);
            w->y_start = w->x_start + diag_at(cq + o->chain_off, ct + o->chain_off, o->n_chain, w->x_start);
            window_verify(x, y, ylen, o->rev, w, ybuf, P->k_cap);
            if (w->err >= 0) o->align_len += w->x_len;
        }
        /* rescue, right-extension pass (Correct.cpp:2655-2744) */
        for (j = o->n_win - 1; j >= 0; j--) {
            orc_win *w = &W[o->first_win + j];
            int k2, next;
            if (w->err < 0) continue;
            next = w->y_beg + w->end_site - w->extra_begin + 1;
            for (k2 = j + 1; k2 < o->n_win && W[o->first_win + k2].err < 0; k2++) {
                orc_win *u = &W[o->first_win + k2], trial = *u;
                if (next >= ylen) break;
                trial.k = (uint8_t)orc_double_thr_p(P, u->k, u->x_len);
                trial.y_start = next;
                if (!window_verify(x, y, ylen, o->rev, &trial, ybuf, P->k_cap)) break;
                if ((trial.x_len + 2 * trial.k - trial.extra_begin - trial.extra_end) + trial.k < trial.x_len) break;
                if (trial.err < 0) break;
                trial.rescued = 1;
                *u = trial;
                o->align_len += u->x_len;
                next = u->y_beg + u->end_site - u->extra_begin + 1;
            }
        }
        /* rescue, left-extension pass (Correct.cpp:2745-2905): a matched window whose left neighbour is unmatched gets its path first
         * -- its real start on y -- and the unmatched windows to its left are tried again, each placed so that it ends right in
         * front of the window to its right, with the doubled threshold and its path at once */
        if (P->left_rescue) for (j = 1; j < o->n_win; j++) {
            orc_win *w = &W[o->first_win + j];
            int k2, total_y_end;
            if (w->err < 0 || W[o->first_win + j - 1].err >= 0) continue;
            if (!w->pad[0]) { window_path(x, y, ylen, o->rev, w, ybuf, cols, tmp, rl, ro); w->pad[0] = 1; }
            if (w->err < 0) continue;
            total_y_end = w->ry_start - 1;
            for (k2 = j - 1; k2 >= 0 && W[o->first_win + k2].err < 0; k2--) {
                orc_win *u = &W[o->first_win + k2], trial = *u;
                trial.k = (uint8_t)orc_double_thr_p(P, u->k, u->x_len);
                if (total_y_end <= 0) break;
                trial.y_start = total_y_end - trial.x_len + 1;
                if (!window_verify(x, y, ylen, o->rev, &trial, ybuf, P->k_cap)) break;
                if ((trial.x_len + 2 * trial.k - trial.extra_begin - trial.extra_end) + trial.k < trial.x_len) break;
                if (trial.err < 0) break;
                window_path(x, y, ylen, o->rev, &trial, ybuf, cols, tmp, rl, ro);
                if (trial.err < 0) break;
                trial.rescued = 1; trial.pad[0] = 1;
                *u = trial;
                o->align_len += u->x_len;
                total_y_end = u->ry_start - 1;
            }
        }
        if (getenv("ORC_DEBUG_WIN")) {
            int dq = -1, dp = -1;
            sscanf(getenv("ORC_DEBUG_WIN"), "%d,%d", &dq, &dp);
            if ((int)o->q == dq && o->x_s <= dp && dp <= o->x_e) {
                fprintf(stderr, "OVL q %u t %u rev %d x [%d,%d] y [%d,%d] nwin %d:", o->q, o->t, o->rev, o->x_s, o->x_e, o->y_s, o->y_e, o->n_win);
                for (j = 0; j < o->n_win; j++) fprintf(stderr, " %d%s", W[o->first_win + j].err, W[o->first_win + j].rescued ? "r" : "");
                fprintf(stderr, "  (win of pos: %d)\n", dp / ORC_WINDOW - o->x_s / ORC_WINDOW);
            }
        }
        /* accept: 0.9 coverage filter, then error rate <= 0.03 with unmatched windows charged in full.
         * NOT RESTATED: non_trim_error_rate (Correct.cpp:725-845) charges an unmatched window that lies beside a matched one less than its
         * length -- it extends the neighbours' alignments into the window from the left and from the right (verify_sub_window :675,
         * Reserve_Banded_BPM_Extension Levenshtein_distance.h:63, doubled threshold) and charges their errors plus the bases neither reaches.
         * In a phased set no overlap's verdict hangs on it (all 204 golden sets agree without it); in an unphased set it decides whether an overlap across a
         * heterozygous indel of 100-200 bases is accepted: round 1 of 5 of the 48 mixed sets (KNOWN_MIXED_READ_DEVIATIONS in the
         * tests; rounds 2 and 3 agree again). */
        o->is_match = 0;
        if (P->partial_charge && (int64_t)(o->x_e - o->x_s + 1) * 9 <= (int64_t)o->align_len * 10) {
            /* the reference has the matched windows' cigars (and the errors / ends they give) before it sums up (Correct.cpp:2920-3003) */
            for (j = 0; j < o->n_win; j++) {
                orc_win *w = &W[o->first_win + j];
                if (w->err >= 0 && !w->pad[0]) { window_path(x, y, ylen, o->rev, w, ybuf, cols, tmp, rl, ro); w->pad[0] = 1; }
            }
        }
        for (j = 0; j < o->n_win; j++) {
            orc_win *w = &W[o->first_win + j];
            tlen += w->x_len;
            if (w->err >= 0) terr += w->err;
            else if (!P->partial_charge) terr += w->x_len;
            else terr = unmatched_charge(x, y, ylen, o->rev, W + o->first_win, o->n_win, j, P, ybuf, terr);
        }
        o->err_sum = (int32_t)terr;
        if ((int64_t)(o->x_e - o->x_s + 1) * 9 <= (int64_t)o->align_len * 10 && terr * 1000 <= tlen * P->accept_err_pm) o->is_match = 1;
        if (!o->is_match) continue;
        for (j = 0; j < o->n_win; j++) {
            orc_win *w = &W[o->first_win + j];
            if (w->err >= 0 && !w->pad[0]) window_path(x, y, ylen, o->rev, w, ybuf, cols, tmp, rl, ro);
        }
    }
    g_fix_boundary = fix_saved_;
    (void)wi;
    *n_win_out = total;
    return W;
}

/* ---------------------------------------------------------------- the consensus of the strings inserted in front of a column
 * build_DAGCon (Correct.cpp:3893-3951): every distinct inserted string is a chain S -> b1 -> ... -> E weighted by the number of
 * overlaps that insert it (in the order the strings first appear); Merge_DAGCon (:3550) goes through the nodes in topological order
 * and merges, per base, the in-nodes of a node that have one out-edge and the out-nodes that have one in-edge (Merge_In_Nodes /
 * Merge_Out_Nodes :3219-3430: from S a prefix trie, from E backwards the unary tails); generate_best_seq_from_nodes (:3804) gives
 * every node the sum of its out-edges as weight (E: of its in-edges) and walks greedily -- forward from S's heaviest out-node when
 * that weighs at least as much as E's heaviest in-node, else backward from that in-node; as E (S) carries the weight of ALL strings
 * the walk ends at the first node where some string ends (starts).  Restated on small arrays, edge lists in insertion order with
 * tombstones as in POA.h:214-490.  Returns the weight of the walk's first node (max_insertion_count) and the string. */
#define DG_MAXN 64        /* bounds of the HIP path (FSV_DG_*): beyond them the most frequent string is inserted */
#define DG_MAXE 128
#define DG_MAXD 8         /* distinct strings */
#define DG_MAXA 8         /* edges on one side of a node */
#define DG_MAXK 64        /* inserted strings at one column */
typedef struct {
    int n_node, n_edge;
    char base[DG_MAXN]; uint8_t alive[DG_MAXN];
    int e_from[DG_MAXE], e_to[DG_MAXE], e_w[DG_MAXE]; uint8_t e_alive[DG_MAXE], e_vis[DG_MAXE];
    int out_n[DG_MAXN], in_n[DG_MAXN];
    int16_t out_e[DG_MAXN][DG_MAXA], in_e[DG_MAXN][DG_MAXA];
    int ok;
} dag_t;

static int dg_node(dag_t *D, char b)
{
    int id = D->n_node;
    if (id >= DG_MAXN) { D->ok = 0; return DG_MAXN - 1; }
    D->base[id] = b; D->alive[id] = 1; D->out_n[id] = D->in_n[id] = 0; D->n_node++;
    return id;
}
static void dg_edge(dag_t *D, int u, int v, int w, int vis)
{
    int e = D->n_edge;
    if (e >= DG_MAXE || D->out_n[u] >= DG_MAXA || D->in_n[v] >= DG_MAXA) { D->ok = 0; return; }
    D->e_from[e] = u; D->e_to[e] = v; D->e_w[e] = w; D->e_alive[e] = 1; D->e_vis[e] = (uint8_t)vis; D->n_edge++;
    D->out_e[u][D->out_n[u]++] = (int16_t)e; D->in_e[v][D->in_n[v]++] = (int16_t)e;
}
static int dg_find(const dag_t *D, int u, int v) /* get_bi_Edge: the in-edges of v in order */
{
    int i;
    for (i = 0; i < D->in_n[v]; i++) { int e = D->in_e[v][i]; if (D->e_alive[e] && D->e_from[e] == u) return e; }
    return -1;
}
static int dg_outdeg(const dag_t *D, int u) { int i, c = 0; for (i = 0; i < D->out_n[u]; i++) c += D->e_alive[D->out_e[u][i]]; return c; }
static int dg_indeg(const dag_t *D, int u) { int i, c = 0; for (i = 0; i < D->in_n[u]; i++) c += D->e_alive[D->in_e[u][i]]; return c; }
static void dg_delete(dag_t *D, int x)
{
    int i;
    D->alive[x] = 0; D->base[x] = 'D';
    for (i = 0; i < D->out_n[x]; i++) D->e_alive[D->out_e[x][i]] = 0;
    for (i = 0; i < D->in_n[x]; i++) D->e_alive[D->in_e[x][i]] = 0;
    D->out_n[x] = D->in_n[x] = 0;
}
static void dg_merge_out(dag_t *D, int cur)
{
    static const char B4[4] = {'A', 'C', 'G', 'T'};
    int bi;
    if (!D->alive[cur] || dg_outdeg(D, cur) == 0) return;
    for (bi = 0; bi < 4; bi++) {
        int flag = 0, weight = 0, cons = -1, i;
        for (i = 0; i < D->out_n[cur]; i++) {
            int e = D->out_e[cur][i], g;
            if (!D->e_alive[e]) continue;
            g = D->e_to[e];
            if (D->base[g] != B4[bi] || dg_indeg(D, g) != 1) continue;
            if (flag == 0) { flag = 1; cons = g; D->e_vis[e] = 1; weight = D->e_w[e]; }
            else {
                int j;
                flag++;
                weight += D->e_w[e];
                for (j = 0; j < D->out_n[g]; j++) {
                    int e2 = D->out_e[g][j], o, e3;
                    if (!D->e_alive[e2]) continue;
                    o = D->e_to[e2];
                    e3 = dg_find(D, cons, o);
                    if (e3 >= 0) { D->e_vis[e3] = 1; D->e_w[e3] += D->e_w[e2]; }
                    else dg_edge(D, cons, o, D->e_w[e2], 1);
                }
                dg_delete(D, g);
            }
        }
        if (flag > 1) { int e = dg_find(D, cur, cons); if (e >= 0) D->e_w[e] = weight; }
        if (flag > 0) dg_merge_out(D, cons);
    }
}
static void dg_merge_in(dag_t *D, int cur)
{
    static const char B4[4] = {'A', 'C', 'G', 'T'};
    int bi;
    if (!D->alive[cur] || dg_indeg(D, cur) == 0) return;
    for (bi = 0; bi < 4; bi++) {
        int flag = 0, weight = 0, cons = -1, i;
        for (i = 0; i < D->in_n[cur]; i++) {
            int e = D->in_e[cur][i], g;
            if (!D->e_alive[e]) continue;
            g = D->e_from[e];
            if (D->base[g] != B4[bi] || dg_outdeg(D, g) != 1) continue;
            if (flag == 0) { flag = 1; cons = g; D->e_vis[e] = 1; weight = D->e_w[e]; }
            else {
                int j;
                flag++;
                weight += D->e_w[e];
                for (j = 0; j < D->in_n[g]; j++) {
                    int e2 = D->in_e[g][j], o, e3;
                    if (!D->e_alive[e2]) continue;
                    o = D->e_from[e2];
                    e3 = dg_find(D, o, cons);
                    if (e3 >= 0) { D->e_vis[e3] = 1; D->e_w[e3] += D->e_w[e2]; }
                    else dg_edge(D, o, cons, D->e_w[e2], 1);
                }
                dg_delete(D, g);
            }
        }
        if (flag > 1) { int e = dg_find(D, cons, cur); if (e >= 0) D->e_w[e] = weight; }
        if (flag > 0) dg_merge_in(D, cons);
    }
}
static int dg_weight(const dag_t *D, int u, int in) /* sum of the out-edges (E: in-edges) */
{
    int i, w = 0;
    if (in) { for (i = 0; i < D->in_n[u]; i++) if (D->e_alive[D->in_e[u][i]]) w += D->e_w[D->in_e[u][i]]; }
    else for (i = 0; i < D->out_n[u]; i++) if (D->e_alive[D->out_e[u][i]]) w += D->e_w[D->out_e[u][i]];
    return w;
}
/* keys[i] = len << 24 | 2-bit bases (len <= INS_MAXLEN), one per overlap in overlap order; out: the consensus string as a key */
static int dagcon_insertion(const uint32_t *keys, int n, uint32_t *out_key)
{
    static __thread dag_t D;
    uint32_t distinct[DG_MAXD]; int cnt[DG_MAXD], nd = 0, i, j, S, E, qh = 0, qt = 0, best_s = -1, best_e = -1, ws = 0, we = 0, cur, L = 0;
    static __thread int queue[4 * DG_MAXE];
    char seq[DG_MAXN];
    *out_key = 0;
    for (i = 0; i < n; i++) {
        for (j = 0; j < nd && distinct[j] != keys[i]; j++) {}
        if (j == nd) { if (nd == DG_MAXD) return -1; distinct[nd] = keys[i]; cnt[nd] = 0; nd++; }
        cnt[j]++;
    }
    D.n_node = D.n_edge = 0; D.ok = 1;
    S = dg_node(&D, 'S'); E = dg_node(&D, 'E');
    for (i = 0; i < nd; i++) {
        int len = (int)(distinct[i] >> 24), last = S;
        for (j = 0; j < len; j++) { int nn = dg_node(&D, "ACGT"[(distinct[i] >> (2 * j)) & 3]); dg_edge(&D, last, nn, cnt[i], 0); last = nn; }
        if (last != S) dg_edge(&D, last, E, cnt[i], 0);
    }
    queue[qt++] = S;
    while (qh < qt && D.ok) {
        cur = queue[qh++];
        dg_merge_in(&D, cur);
        dg_merge_out(&D, cur);
        if (!D.alive[cur]) continue;
        for (i = 0; i < D.out_n[cur]; i++) if (D.e_alive[D.out_e[cur][i]]) D.e_vis[D.out_e[cur][i]] = 1;
        for (i = 0; i < D.out_n[cur]; i++) {
            int e = D.out_e[cur][i], o, all = 1;
            if (!D.e_alive[e]) continue;
            o = D.e_to[e];
            for (j = 0; j < D.in_n[o]; j++) if (D.e_alive[D.in_e[o][j]] && !D.e_vis[D.in_e[o][j]]) { all = 0; break; }
            if (all) { if (qt < 4 * DG_MAXE) queue[qt++] = o; else D.ok = 0; }
        }
    }
    if (!D.ok) return -1;
    for (i = 0; i < D.out_n[S]; i++) { int e = D.out_e[S][i]; if (D.e_alive[e]) { int o = D.e_to[e], w = o == E ? dg_weight(&D, E, 1) : dg_weight(&D, o, 0); if (w > ws) { ws = w; best_s = o; } } }
    for (i = 0; i < D.in_n[E]; i++) { int e = D.in_e[E][i]; if (D.e_alive[e]) { int o = D.e_from[e], w = o == S ? dg_weight(&D, S, 0) : dg_weight(&D, o, 0); if (w > we) { we = w; best_e = o; } } }
    if (ws >= we) {
        cur = best_s;
        while (cur >= 0 && cur != E && L < DG_MAXN) {
            int mx = 0, nx = -1;
            seq[L++] = D.base[cur];
            for (i = 0; i < D.out_n[cur]; i++) { int e = D.out_e[cur][i]; if (D.e_alive[e]) { int o = D.e_to[e], w = o == E ? dg_weight(&D, E, 1) : dg_weight(&D, o, 0); if (w > mx) { mx = w; nx = o; } } }
            cur = nx;
        }
    } else {
        cur = best_e;
        while (cur >= 0 && cur != S && L < DG_MAXN) {
            int mx = 0, nx = -1;
            seq[L++] = D.base[cur];
            for (i = 0; i < D.in_n[cur]; i++) { int e = D.in_e[cur][i]; if (D.e_alive[e]) { int o = D.e_from[e], w = o == S ? dg_weight(&D, S, 0) : dg_weight(&D, o, 0); if (w > mx) { mx = w; nx = o; } } }
            cur = nx;
        }
        for (i = 0; i < L / 2; i++) { char t = seq[i]; seq[i] = seq[L - 1 - i]; seq[L - 1 - i] = t; }
    }
    if (L > INS_MAXLEN) L = INS_MAXLEN;
    *out_key = (uint32_t)L << 24;
    for (i = 0; i < L; i++) *out_key |= (uint32_t)base2(seq[i]) << (2 * i);
    return ws >= we ? ws : we;
}

/* One alignment of a partner read against a stretch of a backbone: the votes it casts (addmatchedSeqToGraph, POA.cpp:309). */
typedef struct { const char *y; int ylen, rev, ry_start, xs, path_len, pend0; const uint8_t *path; uint32_t pend_key; } vote_aln;

typedef struct { int32_t (*cnt)[6]; int32_t *instot; ins_list ins; int32_t (*flg)[4]; } vote_ws;

/* Consensus of the backbone xf[gs, gs+glen) given the alignments (get_seq_from_Graph, Correct.cpp:4010-4129, as a per-column
 * vote); homopolymer tests look at xf[0, xf_len).  Writes the consensus to out, and for every backbone column where its own
 * output base (kept or replaced) sits in out, or -1 when the column was deleted (col_idx may be NULL).  Returns the length. */
static int vote_consensus(const char *xf, int xf_len, int gs, int glen, const vote_aln *A, int nA, vote_ws *V, char *out, int *col_idx, int use_dag)
{
    int i, c, outn = 0;
    memset(V->cnt, 0, sizeof(int32_t[6]) * (ORC_WINDOW + 1));
    V->ins.n = 0;
    memset(V->instot, 0, sizeof(int32_t) * (ORC_WINDOW + 1));
    memset(V->flg, 0, sizeof(int32_t[4]) * (ORC_WINDOW + 1));
    for (i = 0; i < nA; i++) {
        const vote_aln *a = &A[i];
        int xp = a->xs, yp = a->ry_start, p, pend = a->pend0, prev_run = -1, cur_run = -1;
        if (pend && (a->pend_key >> 24)) ins_vote(&V->ins, 0, a->pend_key);
        for (p = 0; p < a->path_len; ) {
            int op = a->path[p];
            if (op != cur_run) { prev_run = cur_run; cur_run = op; }
            if (op == 2) { /* run of y-only bases in front of column xp */
                int L = 0;
                while (p + L < a->path_len && a->path[p + L] == 2) L++;
                if (xp < glen) {
                    pend = 1;
                    if (L <= INS_MAXLEN) {
                        uint32_t key = (uint32_t)L << 24; int b;
                        for (b = 0; b < L; b++) key |= (uint32_t)base2(ybase(a->y, a->ylen, a->rev, yp + b)) << (2 * b);
                        ins_vote(&V->ins, xp, key);
                    }
                }
                yp += L; p += L;
                continue;
            }
            V->cnt[xp][5]++;               /* reads arriving at this column */
            if (pend) { V->instot[xp]++; pend = 0; } /* ... of which after an insertion */
            if (op == 3) V->cnt[xp][4]++;  /* x base without partner: deletion vote */
            else {
                const int yb = base2(ybase(a->y, a->ylen, a->rev, yp));
                V->cnt[xp][yb]++; yp++;
                /* add_mismatchEdge_weight (POA.h:492): last_operation is the previous cigar RUN, so every base of the run that follows an
                 * insertion counts as "after an insertion" on its edge, not only the first */
                if (prev_run == 2) V->flg[xp][yb]++;
            }
            xp++; p++;
        }
    }
    {
        /* get_seq_from_Graph (Correct.cpp:4010-4129) at the node in front of column c, statement by statement */
        for (c = 0; c < glen; c++) {
            const int own = base2(xf[gs + c]);
            const int homo = c > 0 && is_homopolymer_site(xf, xf_len, gs + c - 1);   /* if_is_homopolymer_strict(r_string_site + currentNodeID - 1) */
            int NI = V->instot[c], visit, b;
            for (visit = 0; visit < 2; visit++) {
                int W[4], maxc = -1, type = 0, edge = own, total = 0, order[4], k = 0;
                uint32_t key = 0;
                for (b = 0; b < 4; b++) W[b] = V->cnt[c][b] + (b == own);
                order[k++] = own;
                for (b = 0; b < 4; b++) if (b != own) order[k++] = b;
                for (k = 0; k < 4; k++) {
                    int cw;
                    b = order[k];
                    if (W[b] == 0) continue;
                    cw = NI ? W[b] - V->flg[c][b] : W[b];
                    total += cw;
                    if (cw > maxc) { maxc = cw; type = 0; edge = b; }
                }
                if (NI) {
                    uint32_t ks[DG_MAXK + 1]; int nk = 0, z, mi, z2;
                    total += NI;
                    for (z = 0; z < V->ins.n && nk <= DG_MAXK; z++) if (V->ins.e[z].col == c) ks[nk++] = V->ins.e[z].key;
                    /* hifiasm meets the strings in the order of its overlap list; here they are taken in ascending key order (only
                     * ties between strings of equal weight depend on it), so that the oracle and the HIP path agree */
                    for (z = 1; z < nk; z++) { const uint32_t kv = ks[z]; for (z2 = z; z2 > 0 && ks[z2 - 1] > kv; z2--) ks[z2] = ks[z2 - 1]; ks[z2] = kv; }
                    mi = nk == 0 ? 0 : (nk > DG_MAXK || !use_dag ? -1 : dagcon_insertion(ks, nk, &key));
                    if (mi < 0) mi = ins_winner(&V->ins, c, &key);   /* beyond the DAG's bounds: the most frequent string */
                    if (mi > maxc) { maxc = mi; type = 1; }
                }
                if (V->cnt[c][4]) { total += V->cnt[c][4]; if (V->cnt[c][4] > maxc) { maxc = V->cnt[c][4]; type = 2; } }
                if (maxc * 5 >= total * 3 || (homo && maxc * 1000 >= total * 515)) {
                    if (type == 1) { int L = (int)(key >> 24); for (b = 0; b < L; b++) out[outn++] = "ACGT"[(key >> (2 * b)) & 3]; NI = 0; continue; }
                    if (type == 2) { if (col_idx) col_idx[c] = -1; break; }
                    if (col_idx) col_idx[c] = outn;
                    out[outn++] = "ACGT"[edge];
                    break;
                }
                if (col_idx) col_idx[c] = outn;
                out[outn++] = "ACGT"[own];
                break;
            }
        }
    }
    return outn;
}

#define ORC_BOUNDARY_HALF 187   /* WINDOW_BOUNDARY / 2 (Hash_Table.h:10) */
#define ORC_BOUNDARY_SIDE 25    /* WINDOW_UNCORRECT_SINGLE_SIDE_BOUNDARY (Hash_Table.h:12) */

/* consensus of read q given all accepted overlaps; returns new length, writes into out (cap >= 2*len+64).
 * generate_consensus (Correct.cpp:4731-4808): first the grid windows one by one (window_consensus :4132); then, when
 * P->second_round, every junction between two grid windows again (process_boundary :4453-4728): the 375 bases of the FIRST-round
 * result centred on the junction are the backbone, every overlap that covers the start of the later window is re-aligned to
 * it (threshold doubled once on failure), and the inner 325 bases are replaced by the consensus of those alignments
 * (merge_cigars :4267: from the first to the last kept column inside [25, len - 25)). */
static int correct_read(const readset *R, const orc_asm_params *P, int q, const orc_ovl *ov, int n_ov, const orc_win *W, char *out)
{
    const char *x = R->seq[q];
    int xlen = R->len[q], nwin = (xlen + ORC_WINDOW - 1) / ORC_WINDOW, g, i, outn = 0;
    vote_ws V;
    vote_aln *A;
    int *lb = (int *)malloc(sizeof(int) * (size_t)(nwin + 1));      /* corrected length after every window */
    uint8_t *covered = (uint8_t *)calloc((size_t)nwin + 1, 1);     /* window went through window_consensus */
    /* overlaps of q are contiguous in ov[] (generated q-major) */
    int o0 = 0, o1;
    while (o0 < n_ov && (int)ov[o0].q != q) o0++;
    o1 = o0;
    while (o1 < n_ov && (int)ov[o1].q == q) o1++;
    V.cnt = (int32_t (*)[6])malloc(sizeof(int32_t[6]) * (ORC_WINDOW + 1));
    V.instot = (int32_t *)malloc(sizeof(int32_t) * (ORC_WINDOW + 1));
    V.flg = (int32_t (*)[4])malloc(sizeof(int32_t[4]) * (ORC_WINDOW + 1));
    V.ins.e = NULL; V.ins.n = V.ins.cap = 0;
    A = (vote_aln *)malloc(sizeof(vote_aln) * (size_t)(o1 - o0 + 1));

    for (g = 0; g < nwin; g++) {
        int gs = g * ORC_WINDOW, glen = (gs + ORC_WINDOW <= xlen ? ORC_WINDOW : xlen - gs), cover = 0, nA = 0;
        for (i = o0; i < o1; i++) {
            const orc_ovl *o = &ov[i];
            const orc_win *w;
            int j = g - o->x_s / ORC_WINDOW;
            vote_aln *a;
            if (o->is_match != 1 || j < 0 || j >= o->n_win) continue;
            w = &W[o->first_win + j];
            cover++;                      /* get_available_interval (Correct.cpp:113): every accepted overlap that overlaps the window counts, matched there or not */
            if (w->err < 0) continue;
            a = &A[nA++];
            a->y = R->seq[o->t]; a->ylen = R->len[o->t]; a->rev = o->rev; a->ry_start = w->ry_start; a->xs = w->x_start - gs;
            a->path = w->path; a->path_len = w->path_len; a->pend0 = 0; a->pend_key = 0;
            /* junction with the previous window of the same overlap: y bases skipped by both end-free
             * alignments are an insertion in front of this window's first column */
            if (!P->second_round && j > 0 && W[o->first_win + j - 1].err >= 0) {
                int gap = w->ry_start - W[o->first_win + j - 1].ry_end - 1;
                if (gap > 0 && a->xs == 0) {
                    a->pend0 = 1;
                    if (gap <= INS_MAXLEN) {
                        uint32_t key = (uint32_t)gap << 24; int b;
                        for (b = 0; b < gap; b++) key |= (uint32_t)base2(ybase(a->y, a->ylen, o->rev, w->ry_start - gap + b)) << (2 * b);
                        a->pend_key = key;
                    }
                }
            }
        }
        if (getenv("ORC_DEBUG_VOTE")) {      /* "read,position": the window alignments that vote there, as run-length cigars */
            int dq = -1, dp = -1, z;
            sscanf(getenv("ORC_DEBUG_VOTE"), "%d,%d", &dq, &dp);
            if (dq == q && gs <= dp && dp < gs + glen) for (i = 0; i < nA; i++) {
                fprintf(stderr, "VOTE win %d cover %d aln %d rev %d xs %d ry_start %d:", gs, cover, i, A[i].rev, A[i].xs, A[i].ry_start);
                for (z = 0; z < A[i].path_len; ) { int z2 = z; while (z2 < A[i].path_len && A[i].path[z2] == A[i].path[z]) z2++; fprintf(stderr, " %d%c", z2 - z, "MXID"[A[i].path[z]]); z = z2; }
                fprintf(stderr, "\n");
            }
        }
        if (cover < 3) { /* MIN_COVERAGE_THRESHOLD: copy verbatim */
            memcpy(out + outn, x + gs, (size_t)glen); outn += glen;
        } else {
            outn += vote_consensus(x, xlen, gs, glen, A, nA, &V, out + outn, NULL, P->ins_dag);
            covered[g] = 1;
        }
        lb[g] = outn;
    }
    if (P->second_round && nwin > 1) {
        /* replacements in first-round coordinates, ascending and disjoint */
        int nrep = 0, *rs = (int *)malloc(sizeof(int) * (size_t)nwin * 3), total_extra = 0;
        char **rstr = (char **)malloc(sizeof(char *) * (size_t)nwin);
        orc_win *tw = (orc_win *)malloc(sizeof(orc_win) * (size_t)(o1 - o0 + 1));
        char ybuf[ORC_WINDOW + 2 * ORC_K_WIDE + 8];
        static __thread uint64_t cols[5 * 4 * (ORC_WINDOW + 4)];
        uint8_t tmp[2 * ORC_WINDOW + 4 * ORC_K_WIDE + 16];
        int rl[2 * ORC_WINDOW + 64];
        uint8_t ro[2 * ORC_WINDOW + 64];
        int col_idx[ORC_WINDOW + 1];
        char cons[2 * ORC_WINDOW + 64 + ORC_WINDOW * 16];
        for (g = 1; g < nwin; g++) {
            int gs = g * ORC_WINDOW, LB = lb[g - 1], len_now = lb[g], cws, cwe, blen, nA = 0, terr = 0, k0, sb, eb, c, xs_ = -1, xe_ = -1, cn;
            if (!covered[g] || LB == 0) continue;
            cws = LB - ORC_BOUNDARY_HALF; cwe = LB + ORC_BOUNDARY_HALF - 1;
            if (cws < 0) cws = 0;
            if (cwe >= len_now) cwe = len_now - 1;
            blen = cwe - cws + 1;
            k0 = orc_thr_for_len_p(P, blen);
            for (i = o0; i < o1; i++) {
                const orc_ovl *o = &ov[i];
                const orc_win *w;
                orc_win *t = &tw[nA];
                const char *y = R->seq[o->t];
                int ylen = R->len[o->t], j = g - o->x_s / ORC_WINDOW;
                vote_aln *a;
                if (o->is_match != 1 || j < 0 || j >= o->n_win) continue;
                w = &W[o->first_win + j];
                if (w->err < 0 || w->x_start != gs) continue;
                memset(t, 0, sizeof(*t));
                t->x_start = cws; t->x_len = (int16_t)blen; t->k = (uint8_t)k0; t->y_start = w->ry_start - ORC_BOUNDARY_HALF;
                if (t->y_start < 0) continue;
                if (!window_verify(out, y, ylen, o->rev, t, ybuf, P->k_cap) || t->err < 0) {
                    int k2 = k0 * 2;
                    if (k2 == 0 && blen >= 4) k2 = 1;
                    if (blen >= 300 && k2 < P->k_cap) k2 = P->k_cap;
                    if (k2 > P->k_cap) k2 = P->k_cap;
                    t->k = (uint8_t)k2; t->y_start = w->ry_start - ORC_BOUNDARY_HALF;
                    if (!window_verify(out, y, ylen, o->rev, t, ybuf, P->k_cap) || t->err < 0) continue;
                }
                window_path(out, y, ylen, o->rev, t, ybuf, cols, tmp, rl, ro);
                if (t->err < 0) continue;
                terr += t->err;
                a = &A[nA++];
                a->y = y; a->ylen = ylen; a->rev = o->rev; a->ry_start = t->ry_start; a->xs = 0; a->path = t->path; a->path_len = t->path_len;
                a->pend0 = 0; a->pend_key = 0;
            }
            if (nA < 3 || terr == 0) continue;
            sb = ORC_BOUNDARY_SIDE; eb = blen - 1 - ORC_BOUNDARY_SIDE;
            if (eb <= sb) continue;
            cn = vote_consensus(out, len_now, cws, blen, A, nA, &V, cons, col_idx, P->ins_dag);
            if (cn == blen && !memcmp(cons, out + cws, (size_t)blen)) continue;   /* the new cigar is one run of matches */
            for (c = sb; c < blen && xs_ < 0; c++) if (col_idx[c] >= 0) xs_ = c;
            for (c = eb; c < blen && xe_ < 0; c++) if (col_idx[c] >= 0) xe_ = c;
            if (xs_ < 0 || xe_ < 0) continue;               /* a gap at the end of the stretch: "very likely miscorrection" */
            rs[3 * nrep] = cws + xs_; rs[3 * nrep + 1] = cws + xe_; rs[3 * nrep + 2] = col_idx[xe_] - col_idx[xs_] + 1;
            rstr[nrep] = (char *)malloc((size_t)rs[3 * nrep + 2] + 1);
            memcpy(rstr[nrep], cons + col_idx[xs_], (size_t)rs[3 * nrep + 2]);
            total_extra += rs[3 * nrep + 2];
            nrep++;
        }
        if (nrep) {
            char *fin = (char *)malloc((size_t)outn + (size_t)total_extra + 16);
            int pos = 0, fn = 0, r;
            for (r = 0; r < nrep; r++) {
                memcpy(fin + fn, out + pos, (size_t)(rs[3 * r] - pos)); fn += rs[3 * r] - pos;
                memcpy(fin + fn, rstr[r], (size_t)rs[3 * r + 2]); fn += rs[3 * r + 2];
                pos = rs[3 * r + 1] + 1;
                free(rstr[r]);
            }
            memcpy(fin + fn, out + pos, (size_t)(outn - pos)); fn += outn - pos;
            memcpy(out, fin, (size_t)fn); outn = fn;
            free(fin);
        }
        free(rs); free(rstr); free(tw);
    }
    free(V.cnt); free(V.ins.e); free(V.instot); free(V.flg); free(A); free(lb); free(covered);
    return outn;
}

/* ---------------------------------------------------------------- S7a: haplotype partition of the overlaps of read q
 * partition_overlaps_advance (Correct.cpp:7127-7206) restated: (1) per grid window, columns where at least two overlaps show a
 * mismatch are candidate sites and every overlap covering one leaves an evidence (same base / other base / x base without
 * partner) -- cluster_advance :5585-5738, markSNP_detail :4998, addSNPtohaplotype_details :5247 (window cigars only: the
 * re-aligned junction cigars of calculate_boundary_cigars :2310 are not restated); (2) split_sub_list :5804-5910 keeps a site
 * when one alternative base dominates; (3) generate_haplotypes_DP :6677-6913: sites next to another site are dropped, an overlap
 * that is informative, then not, then informative again is set aside (is_match 4), the longest chains of mutually compatible
 * site vectors are enumerated (Preorder_Merge_Advance_Repeat :6233) and a chain whose two alleles both have support
 * (if_snp_vector_useful :6356) turns the overlaps carrying the other allele into trans overlaps (is_match 2,
 * try_to_remove_reads :6467).  Runs for every read of every set, as hifiasm does -- it has no notion of a phased input. */
#define ORC_SITE_WIN_CAP 255
#define ORC_SITE_RAW_CAP 1024
#define ORC_SITE_READ_CAP 512
typedef struct { int site, occ0, occ1, occ2, homo; int8_t *vec; } snp_site;

static int vec_conflict(const int8_t *a, const int8_t *b, int n)
{
    int i;
    for (i = 0; i < n; i++) if (a[i] != b[i] && (a[i] == 0 || a[i] == 1) && (b[i] == 0 || b[i] == 1)) return 1;
    return 0;
}

typedef struct { snp_site *S; int nS, n; int *maxv, *bt_len, **bt; int *buf; orc_ovl *ov; int *visit; int n_groups; } k7_ctx;

static void k7_group(k7_ctx *C, int plen)
{
    /* process_repeat_snps :6508 for one chain of sites (buf[0..plen), later sites first) */
    int8_t *r = (int8_t *)malloc((size_t)C->n + 1);
    int i, j, occ0 = 0, occ1 = 0, useful = 0;
    memset(r, -1, (size_t)C->n + 1);
    for (j = 0; j < plen; j++) {
        const int8_t *v = C->S[C->buf[j]].vec;
        for (i = 0; i < C->n; i++)
            if (r[i] == -1) { if (v[i] == 0) { occ0++; r[i] = 0; } else if (v[i] == 1) { occ1++; r[i] = 1; } }
    }
    if (occ0 && occ1) {
        double low = (occ0 + occ1) * 0.3;
        if (occ1 >= low && occ0 >= low) useful = 1;
        else if (occ1 >= 5 && occ0 >= 5) useful = 1;
        else if (occ1 >= 3 && occ0 >= 3 && plen >= 2) { /* count_nearby_snps (Correct.h:309): a site 10 or more bases from its chain neighbours */
            int far = 0;
            for (j = 0; j < plen; j++) {
                int cur = C->S[C->buf[j]].site, nearby = 0;
                if (j > 0 && C->S[C->buf[j - 1]].site - cur < 10) nearby = 1;
                if (j + 1 < plen && cur - C->S[C->buf[j + 1]].site < 10) nearby = 1;
                if (!nearby) far++;
            }
            if (far > 0) useful = 1;
        }
    }
    if (useful) for (i = 0; i < C->n; i++) if (r[i] == 1 && C->ov[i].is_match == 1) C->ov[i].is_match = 2;
    free(r);
    C->n_groups++;
}

static void k7_preorder(k7_ctx *C, int id, int plen)
{
    int j;
    C->visit[id] = 1;
    C->buf[plen++] = id;
    if (C->n_groups > 10000) return; /* FSV_K7_GROUP_CAP: the enumeration is exponential in ties; hifiasm has no bound, real data stays far below this */
    if (C->bt_len[id] == 0) { k7_group(C, plen); return; }
    for (j = 0; j < C->bt_len[id]; j++) k7_preorder(C, C->bt[id][j], plen);
}

/* ---- the re-aligned junction cigars of an accepted overlap: calculate_boundary_cigars (Correct.cpp:2310-2530) --------------------
 * For every junction between two matched windows of the overlap whose alignments do not simply meet (bases of y skipped or used
 * twice, or an error within 10 columns of the junction on either side), up to 100 columns on each side of the junction are aligned
 * once more with the doubled threshold (no fix_boundary, no cigar adjustment).  The new cigar is used by the haplotype partition
 * for the ~50 columns on each side of the junction, in place of the two window cigars, unless it has clearly more errors there. */
#define ORC_BC_SIDE 100        /* boundaryLen / 2 */
#define ORC_BC_USELESS 50      /* force_useless_side */
#define ORC_BC_SCAN 10
typedef struct { int avail, x_start, x_end, ry_start, L, R, path_len; uint8_t path[2 * ORC_BC_SIDE + 2 * ORC_K_MAX + 8]; } orc_bcig;

/* scan_cigar (Correct.cpp:1070-1200): errors met while the first (dir 0) / last (dir 1) scan_x columns of x go by; y-only ops count
 * whenever they are met */
static int scan_ops(const uint8_t *path, int plen, int scan_x, int dir)
{
    int x_i = 0, err = 0, p;
    for (p = 0; p < plen; p++) {
        const int op = path[dir ? plen - 1 - p : p];
        if (op == 2) { err++; continue; }
        if (op != 0) err++;
        if (++x_i >= scan_x) return err;
    }
    return err;
}
/* scan_cigar_interval (Correct.cpp:1204-1290): errors over the columns [xb, xe] of x */
static int scan_ops_interval(const uint8_t *path, int plen, int xb, int xe)
{
    int x_i = 0, err = 0, p;
    for (p = 0; p < plen; p++) {
        const int op = path[p];
        if (op == 2) { err++; continue; }
        if (x_i == xb) err = 0;
        x_i++;
        if (op != 0) err++;
        if (x_i == xe + 1) return err;
    }
    return err;
}

static void boundary_cigars(const readset *R, const orc_asm_params *P, const orc_ovl *o, const orc_win *W, orc_bcig *bc)
{
    const char *x = R->seq[o->q], *y = R->seq[o->t];
    const int ylen = R->len[o->t];
    char ybuf[2 * ORC_BC_SIDE + 2 * ORC_K_MAX + 8];
    static __thread uint64_t cols[5 * 4 * (ORC_WINDOW + 4)];
    uint8_t tmp[2 * ORC_WINDOW + 4 * ORC_K_WIDE + 16];
    int rl[2 * ORC_WINDOW + 64];
    uint8_t ro[2 * ORC_WINDOW + 64];
    int i;
    for (i = 0; i + 1 < o->n_win; i++) {
        const orc_win *w0 = &W[o->first_win + i], *w1 = &W[o->first_win + i + 1];
        orc_win t;
        int y_distance, y_start, x_start, x_end, leftLen, rightLen, xLen, thr, L, Rr, m_err, b_err, f_err, o_len;
        bc[i].avail = 0;
        if (w0->err < 0 || w1->err < 0) continue;
        y_distance = w1->ry_start - w0->ry_end - 1;
        if (y_distance == 0 && scan_ops(w0->path, w0->path_len, ORC_BC_SCAN, 1) == 0 && scan_ops(w1->path, w1->path_len, ORC_BC_SCAN, 0) == 0) continue;
        y_start = w0->ry_end; x_start = w0->x_start + w0->x_len - 1;
        leftLen = x_start - w0->x_start; if (y_start < leftLen) leftLen = y_start; if (leftLen > ORC_BC_SIDE) leftLen = ORC_BC_SIDE;
        rightLen = w1->x_start + w1->x_len - x_start; if (ylen - y_start < rightLen) rightLen = ylen - y_start; if (rightLen > ORC_BC_SIDE) rightLen = ORC_BC_SIDE;
        xLen = leftLen + rightLen; x_start -= leftLen; x_end = x_start + xLen - 1; y_start -= leftLen;
        if (xLen <= 0) continue;
        thr = orc_double_thr_p(P, orc_thr_for_len_p(P, xLen), xLen);
        if (thr > ORC_K_MAX) continue;      /* (hifiasm's thresholds: 16 for 200 columns) */
        memset(&t, 0, sizeof t);
        t.x_start = x_start; t.x_len = (int16_t)xLen; t.k = (uint8_t)thr; t.y_start = y_start;
        if (!window_verify(x, y, ylen, o->rev, &t, ybuf, P->k_cap)) continue;
        o_len = xLen + 2 * thr - t.extra_begin - t.extra_end;
        if (o_len < xLen || t.err < 0) continue;
        window_path(x, y, ylen, o->rev, &t, ybuf, cols, tmp, rl, ro);
        if (t.err < 0 || t.path_len > (int)sizeof bc[i].path) continue;
        if (y_distance < 0) y_distance = -y_distance;
        L = Rr = ORC_BC_USELESS;
        if (i == 0 && x_start == W[o->first_win].x_start) L = 0;
        if (i == o->n_win - 2 && x_end == W[o->first_win + o->n_win - 1].x_start + W[o->first_win + o->n_win - 1].x_len - 1) Rr = 0;
        if (leftLen <= L || rightLen <= Rr) continue;
        m_err = scan_ops_interval(t.path, t.path_len, L, xLen - Rr - 1);
        b_err = scan_ops(w0->path, w0->path_len, leftLen - L, 1);
        f_err = scan_ops(w1->path, w1->path_len, rightLen - Rr, 0);
        if (f_err + b_err + y_distance + 1 < m_err) continue;
        bc[i].avail = 1; bc[i].x_start = x_start; bc[i].x_end = x_end; bc[i].ry_start = t.ry_start; bc[i].L = L; bc[i].R = Rr;
        bc[i].path_len = t.path_len; memcpy(bc[i].path, t.path, (size_t)t.path_len);
    }
}

/* the columns [lo, hi] (local to the cigar's first column) of one cigar as evidence per window column: ev[column - gs] = 0 match,
 * 1 + base mismatch, 5 x base without partner (markSNP_detail / addSNPtohaplotype_details, Correct.cpp:4998, :5247) */
static void cigar_evidence(const uint8_t *path, int plen, int x0, int ry0, int lo, int hi, const char *y, int ylen, int rev, int gs, int glen, int8_t *ev)
{
    int x_i = 0, yp = ry0, p;
    if (lo > hi) return;
    for (p = 0; p < plen && x_i <= hi; p++) {
        const int op = path[p];
        if (op == 2) { yp++; continue; }
        if (x_i >= lo) {
            const int c = x0 + x_i - gs;
            if (c >= 0 && c < glen) ev[c] = op == 0 ? 0 : op == 1 ? (int8_t)(1 + base2(ybase(y, ylen, rev, yp))) : 5;
        }
        if (op != 3) yp++;
        x_i++;
    }
}

/* what overlap o shows at the columns of grid window g: the window cigar in the middle, the junction cigars (where available) for
 * the columns beside the two junctions (markSNP_advance / addSNPtohaplotype_advance, Correct.cpp:5054, :5351) */
static void window_evidence(const readset *R, const orc_ovl *o, const orc_win *W, const orc_bcig *bc, int jw, int gs, int glen, int8_t *ev)
{
    const orc_win *w = &W[o->first_win + jw];
    const char *y = R->seq[o->t];
    const int ylen = R->len[o->t], x_total_start = w->x_start, x_length = w->x_len, x_total_end = x_total_start + x_length - 1;
    int cur_beg = 0, cur_end = x_length - 1;
    memset(ev, -1, (size_t)glen);
    if (bc && jw >= 1 && bc[jw - 1].avail) {
        const orc_bcig *b = &bc[jw - 1];
        const int xleft = x_total_start - b->x_start, xright = b->x_end - x_total_start + 1;
        if (xleft > b->L && xright > b->R) {
            cur_beg = xright - b->R;
            cigar_evidence(b->path, b->path_len, b->x_start, b->ry_start, xleft, xleft + (xright - b->R) - 1, y, ylen, o->rev, gs, glen, ev);
        }
    }
    if (bc && jw < o->n_win - 1 && bc[jw].avail) {
        const orc_bcig *b = &bc[jw];
        const int xleft = x_total_end - b->x_start, xright = b->x_end - x_total_end + 1;
        if (xleft > b->L && xright > b->R) {
            cur_end = (x_length - 1) - ((xleft + 1) - b->L);
            cigar_evidence(b->path, b->path_len, b->x_start, b->ry_start, xleft - ((xleft + 1) - b->L) + 1, xleft, y, ylen, o->rev, gs, glen, ev);
        }
    }
    cigar_evidence(w->path, w->path_len, x_total_start, w->ry_start, cur_beg, cur_end, y, ylen, o->rev, gs, glen, ev);
}

static void partition_read(const readset *R, const orc_asm_params *P, int q, orc_ovl *ov, int n_ov, const orc_win *W)
{
    const char *x = R->seq[q];
    int xlen = R->len[q], nwin = (xlen + ORC_WINDOW - 1) / ORC_WINDOW, g, i, j, o0 = 0, o1, n, nS = 0, capS = 0;
    snp_site *S = NULL;
    uint8_t flag[ORC_WINDOW + 1];
    orc_bcig *bcs = NULL;
    int *bc_off = NULL;
    int8_t *evw;
    while (o0 < n_ov && (int)ov[o0].q != q) o0++;
    o1 = o0;
    while (o1 < n_ov && (int)ov[o1].q == q) o1++;
    n = o1 - o0;
    if (n == 0) return;
    if (P->junction_cigars) {      /* calculate_boundary_cigars for every accepted overlap (Correct.cpp:3012) */
        int tot = 0;
        bc_off = (int *)malloc(sizeof(int) * (size_t)(n + 1));
        for (i = o0; i < o1; i++) { bc_off[i - o0] = tot; if (ov[i].is_match == 1 && ov[i].n_win > 1) tot += ov[i].n_win - 1; }
        bc_off[n] = tot;
        bcs = (orc_bcig *)malloc(sizeof(orc_bcig) * (size_t)(tot + 1));
        for (i = o0; i < o1; i++) if (ov[i].is_match == 1 && ov[i].n_win > 1) boundary_cigars(R, P, &ov[i], W, bcs + bc_off[i - o0]);
    }
    evw = (int8_t *)malloc((size_t)n * (ORC_WINDOW + 1));
    for (g = 0; g < nwin; g++) {
        int gs = g * ORC_WINDOW, glen = (gs + ORC_WINDOW <= xlen ? ORC_WINDOW : xlen - gs), c, any = 0, nS_win = nS;
        memset(flag, 0, sizeof flag);
        for (i = o0; i < o1; i++) { /* markSNP_advance: mismatch columns; what every overlap shows at every column of the window */
            const orc_ovl *o = &ov[i];
            int8_t *e = evw + (size_t)(i - o0) * (ORC_WINDOW + 1);
            int jw = g - o->x_s / ORC_WINDOW;
            memset(e, -1, (size_t)glen);
            if (o->is_match != 1 || jw < 0 || jw >= o->n_win) continue;
            if (W[o->first_win + jw].err < 0) continue;
            window_evidence(R, o, W, bcs && o->n_win > 1 ? bcs + bc_off[i - o0] : NULL, jw, gs, glen, e);
            for (c = 0; c < glen; c++) if (e[c] >= 1 && e[c] <= 4 && flag[c] < 127) { flag[c]++; any = 1; }
        }
        if (!any) continue;
        for (c = 0; c < glen; c++) {
            int occ0 = 0, occ1 = 0, occ2 = 0, oa[4] = {0, 0, 0, 0}, mx, mi, b, total;
            int8_t *ev;
            if (flag[c] <= 1) continue;
            ev = (int8_t *)malloc((size_t)n);     /* -1 none, 0 same, 1..4 other base + 1, 5 gap */
            for (i = 0; i < n; i++) { /* addSNPtohaplotype_advance at column c */
                ev[i] = evw[(size_t)i * (ORC_WINDOW + 1) + c];
                if (ev[i] == 0) occ0++;
                else if (ev[i] >= 1 && ev[i] <= 4) { oa[ev[i] - 1]++; occ1++; }
                else if (ev[i] == 5) occ2++;
            }
            /* split_sub_list */
            mx = occ2; mi = -1;
            for (b = 0; b < 4; b++) if (oa[b] > mx) { mx = oa[b]; mi = b; }
            total = occ0 + occ1 + occ2;
            if (occ0 == 0 || occ1 == 0 || mi < 0 || mx <= 1) { free(ev); continue; }
            for (b = 0; b < 4; b++) if (oa[b] == mx && b != mi) mi = -2;
            if (mi < 0) { free(ev); continue; }
            if ((double)(occ0 + 1 + mx) / (double)(total + 1) < 0.95) { free(ev); continue; }
            if ((double)mx / (double)(total + 1 - (occ0 + 1)) < 0.70) { free(ev); continue; }
            if (nS == capS) { capS = capS ? capS * 2 : 16; S = (snp_site *)realloc(S, sizeof(snp_site) * (size_t)capS); }
            S[nS].site = gs + c; S[nS].occ0 = S[nS].occ1 = S[nS].occ2 = 0; S[nS].homo = is_homopolymer_site(x, xlen, gs + c);
            for (i = 0; i < n; i++) { /* InsertSNPVector */
                if (ev[i] == 0) S[nS].occ0++;
                else if (ev[i] == 1 + mi) { ev[i] = 1; S[nS].occ1++; }
                else if (ev[i] > 0) { ev[i] = 2; S[nS].occ2++; }
            }
            S[nS].vec = ev;
            if (getenv("ORC_DEBUG_K7") && atoi(getenv("ORC_DEBUG_K7")) == q) {
                fprintf(stderr, "K7SITE site %d occ0 %d occ1 %d occ2 %d max %d base %c:", gs + c, occ0, occ1, occ2, mx, "ACGT"[mi]);
                for (i = 0; i < n; i++) if (ev[i] >= 0) fprintf(stderr, " t%u:%d", ov[o0 + i].t, ev[i]);
                fprintf(stderr, "\n");
            }
            nS++;
        }
        /* bounds of the HIP path (FSV_SITE_WIN_CAP, FSV_SITE_RAW_CAP, FSV_SITE_READ_CAP; hifiasm has none): a window with more than 255 kept sites
         * contributes none, a read with more than 1 024 (512 once the sites beside another site are gone) is not partitioned */
        if (nS - nS_win > ORC_SITE_WIN_CAP) { while (nS > nS_win) free(S[--nS].vec); }
    }
    free(evw); free(bcs); free(bc_off);
    if (nS > ORC_SITE_RAW_CAP) { for (j = 0; j < nS; j++) free(S[j].vec); free(S); return; }
    if (nS == 0) { free(S); return; }
    /* generate_haplotypes_DP: a site directly beside another one is dropped */
    if (nS > 1) {
        int m = 0;
        uint8_t *keep = (uint8_t *)calloc((size_t)nS, 1);
        for (j = 0; j < nS; j++) {
            int left = j > 0 && S[j].site == S[j - 1].site + 1, right = j + 1 < nS && S[j].site + 1 == S[j + 1].site;
            keep[j] = !(left || right);
        }
        for (j = 0; j < nS; j++) { if (keep[j]) S[m++] = S[j]; else free(S[j].vec); }
        free(keep);
        nS = m;
    }
    if (nS > ORC_SITE_READ_CAP) { for (j = 0; j < nS; j++) free(S[j].vec); free(S); return; }
    for (i = 0; i < n; i++) { /* informative, not informative, informative again: set aside */
        int st = -1;
        if (ov[o0 + i].is_match != 1) continue;
        for (j = 0; j < nS; j++) {
            int inf = S[j].vec[i] == 0 || S[j].vec[i] == 1;
            if (st == -1) { if (inf) st = 0; }
            else if (st == 0) { if (!inf) st = 2; }
            else if (st == 2) { if (inf) { st = 3; break; } }
        }
        if (st == 3) {
            for (j = 0; j < nS; j++) {
                if (S[j].vec[i] == 0) { S[j].occ0--; S[j].occ2++; }
                else if (S[j].vec[i] == 1) { S[j].occ1--; S[j].occ2++; }
                else if (S[j].vec[i] != 2) S[j].occ2++;
                S[j].vec[i] = 2;
            }
            ov[o0 + i].is_match = 4;
        }
    }
    if (nS > 0) {
        k7_ctx C;
        int *order = (int *)malloc(sizeof(int) * (size_t)nS);
        C.S = S; C.nS = nS; C.n = n; C.ov = ov + o0; C.n_groups = 0;
        C.maxv = (int *)malloc(sizeof(int) * (size_t)nS); C.bt_len = (int *)calloc((size_t)nS, sizeof(int));
        C.bt = (int **)malloc(sizeof(int *) * (size_t)nS); C.buf = (int *)malloc(sizeof(int) * (size_t)nS + 8);
        C.visit = (int *)calloc((size_t)nS, sizeof(int));
        for (i = 0; i < nS; i++) {
            int eq = 0;
            C.maxv[i] = 1;
            C.bt[i] = (int *)malloc(sizeof(int) * (size_t)(i + 1));
            for (j = 0; j < i; j++) {
                if (vec_conflict(S[i].vec, S[j].vec, n)) continue;
                if (C.maxv[i] < C.maxv[j] + 1) { C.maxv[i] = C.maxv[j] + 1; C.bt[i][0] = j; eq = 1; }
                else if (C.maxv[i] == C.maxv[j] + 1) C.bt[i][eq++] = j;
            }
            C.bt_len[i] = eq;
        }
        for (i = 0; i < nS; i++) order[i] = i; /* by chain length, longest first (stable, as glibc's qsort is for arrays this small) */
        for (i = 1; i < nS; i++) { int v = order[i]; for (j = i; j > 0 && C.maxv[order[j - 1]] < C.maxv[v]; j--) order[j] = order[j - 1]; order[j] = v; }
        for (i = 0; i < nS; i++) if (!C.visit[order[i]]) k7_preorder(&C, order[i], 0);
        for (i = 0; i < nS; i++) free(C.bt[i]);
        free(C.bt); free(C.maxv); free(C.bt_len); free(C.buf); free(C.visit); free(order);
    }
    for (j = 0; j < nS; j++) free(S[j].vec);
    free(S);
}

/* one correction round: R -> corrected reads (new buffers); returns total windows examined */
static void correction_round(readset *R, const orc_asm_params *P, int w, int do_rc, orc_ovl **accepted, int *n_accepted)
{
    orc_mz **uq; int *nuq, n_ov, n_win, q;
    orc_ovl *ov; int32_t *cq, *ct; orc_win *W;
    char **nseq = (char **)malloc(sizeof(char *) * (size_t)R->n);
    int *nlen = (int *)malloc(sizeof(int) * (size_t)R->n);
    sketch_set(R, P, w, &uq, &nuq);
    g_both_ways = getenv("ORC_EC_BOTH_WAYS") ? 1 : 0;   /* experiment switch: chain the correction rounds' overlaps from both sides too (changes none of the golden sets) */
    collect_overlaps(R, P, P->bw_ec, uq, nuq, &ov, &cq, &ct, &n_ov);
    g_both_ways = 0;
    W = align_overlaps(R, P, ov, n_ov, cq, ct, &n_win);
    if (P->partition) for (q = 0; q < R->n; q++) partition_read(R, P, q, ov, n_ov, W); /* every read of every set, phased or not, as hifiasm */
    for (q = 0; q < R->n; q++) {
        nseq[q] = (char *)malloc((size_t)R->len[q] * 2 + 64 + (size_t)ORC_WINDOW * 16);
        nlen[q] = correct_read(R, P, q, ov, n_ov, W, nseq[q]);
        if (do_rc) revcomp_inplace(nseq[q], nlen[q]);
    }
    for (q = 0; q < R->n; q++) { free(R->seq[q]); R->seq[q] = nseq[q]; R->len[q] = nlen[q]; }
    if (accepted) { /* the overlaps this round verified (coordinates on the reads as they were before the round) */
        int i, m = 0;
        for (i = 0; i < n_ov; i++) if (ov[i].is_match == 1) ov[m++] = ov[i];
        *accepted = ov; *n_accepted = m; ov = NULL;
    }
    free(nseq); free(nlen); free(W); free(ov); free(cq); free(ct);
    free_sketch(R, uq, nuq);
}

/* ---------------------------------------------------------------- S9 + S10 */
typedef struct { int to, to_rev, ovl; } arc_t; /* best successor of an oriented read */

void orc_asm_default_params(orc_asm_params *P)
{
    P->k = 51; P->w = 51; P->hpc = 1; P->n_rounds = 3; P->min_ovlp = 1; P->min_anchors = 1; P->lookback = 64;
    P->bw_ec = 20; P->bw_final = 0; P->min_contig_reads = 4; P->partition = 1;
    P->win_rate_pm = 40; P->k_cap = ORC_K_MAX; P->accept_err_pm = 30; P->bw_rechain = 1; P->w_later = 0; P->second_round = 1; P->ins_dag = 1;
    P->min_anchors_final = 1; P->min_ovlp_final = 1; P->graph_layout = 1;
    P->junction_cigars = 1;
    P->fix_boundary = 1;
    P->partial_charge = getenv("ORC_PARTIAL_CHARGE") ? 1 : 0;   /* (the env switch: to run any test with it) */
    P->left_rescue = 1;   /* recalcate_window_advance's left pass (Correct.cpp:2745-2905); the HIP path: k_left_rescue */
}

/* Overlaps of the corrected reads for the layout (worker_ov_final, Assembly.cpp:1284-1306): exact ones (update_exact_overlaps),
 * and inexact ones that the last correction round had verified for the same ordered pair and strand with about the same
 * coordinates (update_overlaps: both ends of either read within 10 % of the longer span).  Pairs without an exact overlap are
 * re-chained with hifiasm's final bandwidth (0.001) to get gapped coordinates.  Returns the hits in ov[]. */
static int final_overlaps(const readset *R, const orc_asm_params *P0, const orc_ovl *prev, int n_prev, orc_ovl **out)
{
    orc_mz **uq; int *nuq, n_ov, i, m = 0;
    orc_ovl *ov; int32_t *cq, *ct;
    orc_asm_params Pf = *P0;
    const orc_asm_params *P = &Pf;
    /* hifiasm's final pass keeps every pair that shares a minimizer on a strand, whatever the length of the overlap */
    Pf.min_anchors = P0->min_anchors_final > 0 ? P0->min_anchors_final : P0->min_anchors;
    Pf.min_ovlp = P0->min_ovlp_final > 0 ? P0->min_ovlp_final : P0->min_ovlp;
    sketch_set(R, P, P->w_later > 0 ? P->w_later : P->w, &uq, &nuq);
    collect_overlaps(R, P, P->bw_final, uq, nuq, &ov, &cq, &ct, &n_ov);
    for (i = 0; i < n_ov; i++) {
        orc_ovl *o = &ov[i];
        const char *x = R->seq[o->q], *y = R->seq[o->t];
        int ylen = R->len[o->t], L = o->x_e - o->x_s + 1, p, same = (L == o->y_e - o->y_s + 1);
        for (p = 0; same && p < L; p++) if (x[o->x_s + p] != ybase(y, ylen, o->rev, o->y_s + p)) same = 0;
        o->exact = (uint8_t)same;
        if (same) ov[m++] = *o;
    }
    if (n_prev > 0) {
        orc_ovl *ov2; int32_t *cq2, *ct2; int n2, j;
        int32_t *slot = (int32_t *)malloc(sizeof(int32_t) * (size_t)R->n * R->n);
        uint8_t *has = (uint8_t *)calloc((size_t)R->n * R->n, 1);
        for (i = 0; i < R->n * R->n; i++) slot[i] = -1;
        for (i = 0; i < n_prev; i++) slot[(size_t)prev[i].q * R->n + prev[i].t] = i;
        for (i = 0; i < m; i++) has[(size_t)ov[i].q * R->n + ov[i].t] = 1;
        g_both_ways = getenv("ORC_ONE_WAY") ? 0 : 1;
        collect_overlaps(R, P, P->bw_rechain, uq, nuq, &ov2, &cq2, &ct2, &n2);
        g_both_ways = 0;
        ov = (orc_ovl *)realloc(ov, sizeof(orc_ovl) * (size_t)(m + n2 + 1));
        for (j = 0; j < n2; j++) {
            const orc_ovl *o = &ov2[j];
            const size_t key = (size_t)o->q * R->n + o->t;
            const orc_ovl *pv;
            int lx, ly, L;
            if (has[key] || slot[key] < 0) continue;
            pv = &prev[slot[key]];
            if (pv->rev != o->rev) continue;
            lx = pv->x_e - pv->x_s + 1; ly = pv->y_e - pv->y_s + 1;
            L = (lx > ly ? lx : ly) / 10;
            if ((abs(o->x_s - pv->x_s) < L && abs(o->x_e - pv->x_e) < L) || (abs(o->y_s - pv->y_s) < L && abs(o->y_e - pv->y_e) < L)) {
                ov[m] = *o; ov[m].exact = 0; m++;
            }
        }
        free(ov2); free(cq2); free(ct2); free(has); free(slot);
    }
    free(cq); free(ct);
    free_sketch(R, uq, nuq);
    *out = ov;
    return m;
}

/* Layout.  Oriented node v = 2*read + strand.  A hit (q fwd, t strand rev) with x_e == lenq-1, y_s == 0 is the arc
 * (q,+) -> (t,rev) and, complemented, (t,!rev) -> (q,-).  Contained reads are dropped first (ma_hit_contained),
 * every node keeps its longest out-arc (what transitive reduction leaves on error-free linear data), and arcs
 * that are mutually best are walked into unitigs. */
int orc_layout(const int *len, int n, const orc_ovl *hit, int n_hit, int min_reads, int32_t *piece_read, uint8_t *piece_rev,
               int32_t *piece_len, int32_t *contig_first, int piece_cap, int contig_cap)
{
    uint8_t *contained = (uint8_t *)calloc((size_t)n, 1), *used = (uint8_t *)calloc((size_t)n, 1);
    int32_t *succ = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * 2), *sovl = (int32_t *)calloc((size_t)n * 2, sizeof(int32_t));
    int32_t *pred = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * 2);
    int i, v, n_piece = 0, n_contig = 0;
    /* ma_hit2arc (Overlaps.h:178-246) from the query's side of every hit; the mirrored hit supplies the other side.
     * tl5 / tl3 = overhang of the target in front of / behind the overlap, on the query's strand (y is strand-corrected). */
#define HIT_GEOM(h) const int ql = len[(h)->q], tl = len[(h)->t], qs = (h)->x_s, qe = (h)->x_e + 1, tl5 = (h)->y_s, tl3 = tl - ((h)->y_e + 1); \
                    const int ext5 = qs < tl5 ? qs : tl5, ext3 = ql - qe < tl3 ? ql - qe : tl3, tspan = (h)->y_e + 1 - (h)->y_s; \
                    const int internal = ext5 > 1000 || ext3 > 1000 || (qe - qs) < (qe - qs + ext5 + ext3) * 0.8f || tspan < (tspan + ext5 + ext3) * 0.8f
    for (i = 0; i < n_hit; i++) {
        const orc_ovl *h = &hit[i];
        HIT_GEOM(h);
        if (internal) continue;
        if (qs <= tl5 && ql - qe <= tl3) {                 /* MA_HT_QCONT */
            if (qs >= tl5 && ql - qe >= tl3) { if (h->q > h->t) contained[h->q] = 1; } /* mutual: keep the lower index */
            else contained[h->q] = 1;
        }
    }
    for (v = 0; v < 2 * n; v++) { succ[v] = -1; pred[v] = -1; sovl[v] = 0x7fffffff; }
    for (i = 0; i < n_hit; i++) {
        const orc_ovl *h = &hit[i];
        int from, to, l;
        HIT_GEOM(h);
        if (internal || contained[h->q] || contained[h->t]) continue;
        if ((qs <= tl5 && ql - qe <= tl3) || (qs >= tl5 && ql - qe >= tl3)) continue; /* containments */
        if (qe - qs + ext5 + ext3 < 50 || tspan + ext5 + ext3 < 50) continue;         /* MA_HT_SHORT_OVLP */
        if (qs > tl5) { from = 2 * (int)h->q; to = 2 * (int)h->t + h->rev; l = qs - tl5; }              /* (q,+) -> (t,rev) */
        else { from = 2 * (int)h->q + 1; to = 2 * (int)h->t + !h->rev; l = (ql - qe) - tl3; }           /* (q,-) -> (t,!rev) */
        /* every node keeps its nearest successor: the smallest node length = the longest overlap */
        if (l < sovl[from] || (l == sovl[from] && succ[from] >= 0 && to < succ[from])) { succ[from] = to; sovl[from] = l; }
    }
#undef HIT_GEOM
    /* keep mutually-best arcs only: v -> w is kept when the best out-arc of ~w is ~v */
    for (v = 0; v < 2 * n; v++) {
        int w = succ[v];
        if (w >= 0 && succ[w ^ 1] != (v ^ 1)) succ[v] = -1;
    }
    if (getenv("ORC_DEBUG_LAYOUT")) {
        for (i = 0; i < n; i++) fprintf(stderr, "read %d len %d contained %d succ+ %d(%d) succ- %d(%d)\n", i, len[i], contained[i], succ[2*i], sovl[2*i], succ[2*i+1], sovl[2*i+1]);
    }
    for (v = 0; v < 2 * n; v++) if (succ[v] >= 0) pred[succ[v]] = v;
    /* ma_ug_gen (Overlaps.cpp:7759): vertices in increasing order; the unitig through the first unvisited one is emitted in
     * that vertex's direction, from its start (found by walking the in-arcs back).  So the lowest-numbered read of a chain sits
     * on its forward strand -- the two directions spell reverse complements only while every overlap is exact. */
    for (v = 0; v < 2 * n; v++) {
        int r = v >> 1, cnt = 0, w, first = n_piece, start = v, steps = 0;
        if (contained[r] || used[r]) continue;
        while (pred[start] >= 0 && !used[pred[start] >> 1] && steps < 2 * n) { start = pred[start]; steps++; if (start == v) break; }
        for (w = start; w >= 0 && !used[w >> 1] && cnt <= 2 * n; w = succ[w]) { cnt++; if (succ[w] == start) break; }
        if (cnt < min_reads) continue;
        if (n_contig >= contig_cap || n_piece + cnt > piece_cap) break;
        for (w = start; w >= 0 && !used[w >> 1]; w = succ[w]) {
            used[w >> 1] = 1;
            piece_read[n_piece] = w >> 1; piece_rev[n_piece] = (uint8_t)(w & 1);
            piece_len[n_piece] = (succ[w] >= 0 && !used[succ[w] >> 1]) ? sovl[w] : len[w >> 1];
            n_piece++;
        }
        contig_first[n_contig++] = first;
    }
    /* no fall-back: hifiasm's asg_cut_tip (Overlaps.cpp:4666-4709, max_short_tip = 3) deletes every dead-end chain of
     * fewer than four reads, a lone uncontained read included, and then writes no contig at all for the set */
    contig_first[n_contig] = n_piece;
    free(contained); free(used); free(succ); free(sovl); free(pred);
    return n_contig;
}

int orc_assemble(const char *seqs, const uint64_t *seq_off, int n_reads, const orc_asm_params *P,
                 char *contigs, uint64_t contigs_cap, uint64_t *contig_off, int contig_cap, int *n_contigs,
                 char *corrected, uint64_t corrected_cap, uint64_t *corrected_off)
{
    readset R;
    orc_ovl *hits, *prev = NULL; int n_hit, n_prev = 0, r, i, c, nc;
    int32_t *piece_read, *piece_len, *contig_first; uint8_t *piece_rev;
    uint64_t used = 0;
    R.n = n_reads;
    R.seq = (char **)malloc(sizeof(char *) * (size_t)n_reads);
    R.len = (int *)malloc(sizeof(int) * (size_t)n_reads);
    for (r = 0; r < n_reads; r++) {
        R.len[r] = (int)(seq_off[r + 1] - seq_off[r]);
        R.seq[r] = (char *)malloc((size_t)R.len[r] + 1);
        for (i = 0; i < R.len[r]; i++) { char ch = seqs[seq_off[r] + i]; R.seq[r][i] = (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') ? ch : 'A'; }
    }
    for (i = 0; i < P->n_rounds; i++) {
        const int last = i + 1 == P->n_rounds;
        correction_round(&R, P, (i > 0 && P->w_later > 0) ? P->w_later : P->w, !last, last ? &prev : NULL, last ? &n_prev : NULL);
    }
    if (corrected && corrected_off) {
        uint64_t u = 0;
        for (r = 0; r < n_reads; r++) {
            corrected_off[r] = u;
            if (u + (uint64_t)R.len[r] <= corrected_cap) memcpy(corrected + u, R.seq[r], (size_t)R.len[r]);
            u += (uint64_t)R.len[r];
        }
        corrected_off[n_reads] = u;
    }
    n_hit = final_overlaps(&R, P, prev, n_prev, &hits);
    free(prev);
    if (getenv("ORC_DEBUG_HITS")) {
        for (i = 0; i < n_hit; i++) fprintf(stderr, "HIT %u %u %d %d %d %d %d %d\n", hits[i].q, hits[i].t, hits[i].x_s, hits[i].x_e + 1,
            hits[i].rev ? R.len[hits[i].t] - 1 - hits[i].y_e : hits[i].y_s, hits[i].rev ? R.len[hits[i].t] - hits[i].y_s : hits[i].y_e + 1, hits[i].rev, hits[i].exact);
    }
    piece_read = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_reads + 1));
    piece_len = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_reads + 1));
    piece_rev = (uint8_t *)malloc((size_t)n_reads + 1);
    contig_first = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_reads + 2));
    if (P->graph_layout)
        nc = orc_layout_graph((const char *const *)R.seq, R.len, n_reads, hits, n_hit, P->min_contig_reads, piece_read, piece_rev, piece_len, contig_first,
                              n_reads, contig_cap < n_reads ? contig_cap : n_reads);
    else
        nc = orc_layout(R.len, n_reads, hits, n_hit, P->min_contig_reads, piece_read, piece_rev, piece_len, contig_first, n_reads, contig_cap < n_reads ? contig_cap : n_reads);
    for (c = 0; c < nc; c++) {
        contig_off[c] = used;
        for (i = contig_first[c]; i < contig_first[c + 1]; i++) {
            int rd = piece_read[i], L = piece_len[i], p;
            if (used + (uint64_t)L > contigs_cap) { nc = c; goto done; }
            for (p = 0; p < L; p++) contigs[used + p] = ybase(R.seq[rd], R.len[rd], piece_rev[i], p);
            used += (uint64_t)L;
        }
    }
done:
    contig_off[nc] = used;
    *n_contigs = nc;
    for (r = 0; r < n_reads; r++) free(R.seq[r]);
    free(R.seq); free(R.len); free(hits); free(piece_read); free(piece_len); free(piece_rev); free(contig_first);
    return 0;
}
