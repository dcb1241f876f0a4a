/* oracle/layout.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of what hifiasm-0.14 does between its final overlaps and the contig sequence, for the read sets of one
 * region (tens to hundreds of reads of one haplotype):
 *   detect_chimeric_reads        Overlaps.cpp:1698-1784 (collect_sides :1493, collect_contain :1516)
 *   ma_hit_cut / ma_hit_flt      :1785, :1132        (overlaps below 50 bases, internal matches)
 *   ma_hit_contained_advance     :1031-1131          (containment, in read order: a read already removed contains nobody)
 *   ma_sg_gen, ma_hit2arc        :2152, Overlaps.h:178-246
 *   asg_arc_del_trans, asg_symm  :4531-4664, :342    (transitive reduction with fuzz 1000)
 *   asg_cut_tip                  :4666-4720          (dead ends of up to three reads)
 *   ma_ug_gen                    :7759-7900          (unitigs, in vertex order)
 *   polish_unitig                :8480-8560          (a read joined by an inexact overlap is skipped when its neighbours overlap exactly)
 *   polish_unitig_advance        :8893-8960, get_consensus_rate :8800-8890 (the same from the reads' base-level agreement)
 *   ma_ug_seq                    :8962-9034          (contig = the reads' prefixes)
 * Traced with the reference's own code (oracle/ref_graph_trace.cpp -> oracle/_ref/hifiasm_trace): on read sets of this kind none of clean_graph's other steps
 * (Overlaps.cpp:27087-27350: the four cleaning rounds, bubble popping, the rescue passes) changes the graph, so they are not
 * restated; the complex branch of detect_chimeric_reads (a read whose left and right overlaps meet in fewer than 0.2 % of its
 * length AND some spanning overlap fails a window check there) is taken as "not chimeric": final overlaps are exact or were
 * verified window by window in the last correction round.  clean_weak_ma_hit_t is a no-op on the hit lists built here.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

typedef struct { int32_t qn, tn, qs, qe, ts, te; uint8_t rev, el, del, pad; } lhit;   /* ma_hit_t: qe / te exclusive, target in its own forward coordinates */
typedef struct { uint32_t u, v; int32_t len, ol; uint8_t el, del; } larc;            /* asg_arc_t: u -> v, len = node length, ol = overlap length */
enum { HT_INT = -1, HT_QCONT = -2, HT_TCONT = -3, HT_SHORT = -4 };
#define MAX_HANG 1000
#define INT_FRAC 0.8f
#define MIN_OVLP 50
#define GAP_FUZZ 1000
#define MAX_SHORT_TIP 3

typedef struct {
    int n;                 /* reads */
    const int *len;
    const char *const *seq;
    lhit *h; int n_h;      /* all hits; src_first[q] .. src_first[q+1] are read q's, sorted by target */
    int *src_first;
    uint8_t *rdel;         /* coverage_cut[].del */
    larc *arc; int n_arc;  /* sorted by (u, len) */
    int *a_first;          /* 2n + 1 */
    uint8_t *sdel;         /* g->seq[].del */
} lay_t;

static int hit2arc(const lhit *h, int ql, int tl, larc *p)
{
    int32_t tl5, tl3, ext5, ext3, qs = h->qs;
    uint32_t u, v; int32_t l;
    if (h->rev) { tl5 = tl - h->te; tl3 = h->ts; } else { tl5 = h->ts; tl3 = tl - h->te; }
    ext5 = qs < tl5 ? qs : tl5;
    ext3 = ql - h->qe < tl3 ? ql - h->qe : tl3;
    if (ext5 > MAX_HANG || ext3 > MAX_HANG || h->qe - qs < (h->qe - qs + ext5 + ext3) * INT_FRAC || h->te - h->ts < (h->te - h->ts + ext5 + ext3) * INT_FRAC)
        return HT_INT;
    if (qs <= tl5 && ql - h->qe <= tl3) return HT_QCONT;
    else if (qs >= tl5 && ql - h->qe >= tl3) return HT_TCONT;
    else if (qs > tl5) { u = 0; v = !!h->rev; l = qs - tl5; }
    else { u = 1; v = !h->rev; l = (ql - h->qe) - tl3; }
    if (h->qe - qs + ext5 + ext3 < MIN_OVLP || h->te - h->ts + ext5 + ext3 < MIN_OVLP) return HT_SHORT;
    u |= (uint32_t)h->qn << 1; v |= (uint32_t)h->tn << 1;
    p->u = u; p->v = v; p->len = l; p->ol = ql - l; p->el = h->el; p->del = 0;
    return (int)l;
}

static lhit *find_hit(lay_t *L, int qn, int tn)
{
    int i;
    for (i = L->src_first[qn]; i < L->src_first[qn + 1]; i++) if (L->h[i].tn == tn) return &L->h[i];
    return NULL;
}
static void delete_single_edge(lay_t *L, int qn, int tn) { lhit *t = find_hit(L, qn, tn); if (t) t->del = 1; }
static void delete_all_edges(lay_t *L, int qn)
{
    int i;
    for (i = L->src_first[qn]; i < L->src_first[qn + 1]; i++) { L->h[i].del = 1; delete_single_edge(L, L->h[i].tn, qn); }
    L->rdel[qn] = 1;
}

/* normalize_ma_hit_t_single_side_advance (Overlaps.cpp:450-515): the two directions of a pair are made one overlap -- the direction
 * with the longer query interval stands and the other becomes its mirror image (equal lengths: the lower read's stands); a hit
 * without a partner is deleted.  The final pass's gapped re-chain (oracle/asm.c: collect_overlaps, both ways) chains either direction
 * from its own side, and with an indel budget the two can differ by the bases of an indel near a read end. */
static void normalize(lay_t *L)
{
    int i, j;
    for (i = 0; i < L->n; i++)
        for (j = L->src_first[i]; j < L->src_first[i + 1]; j++) {
            lhit *h = &L->h[j], *r = find_hit(L, h->tn, h->qn);
            if (r) {
                const int is_del = h->del || r->del, q0 = h->qe - h->qs, q1 = r->qe - r->qs;
                if ((q0 == q1 && h->qn < h->tn) || q0 > q1) { r->qs = h->ts; r->qe = h->te; r->ts = h->qs; r->te = h->qe; r->rev = h->rev; r->el = h->el; }
                h->del = r->del = (uint8_t)is_del;
            } else h->del = 1;       /* (hifiasm adds the mirror image and deletes both) */
        }
}

/* detect_chimeric_reads: a read whose overlaps from the left end and from the right end do not reach each other */
static void chimeric(lay_t *L)
{
    const float shift_rate = (float)(0.001f * 2.0);
    int i, j;
    for (i = 0; i < L->n; i++) {
        const int64_t rl = L->len[i];
        int64_t ls = rl, le = 0, rs = rl, re = 0, nle, nrs;
        for (j = L->src_first[i]; j < L->src_first[i + 1]; j++) {
            const lhit *h = &L->h[j];
            if (h->del) continue;
            if (h->qs == 0) { if (h->qs < ls) ls = h->qs; if (h->qe > le) le = h->qe; }
            if (h->qe == rl) { if (h->qs < rs) rs = h->qs; if (h->qe > re) re = h->qe; }
        }
        if (ls == rl || rs == rl) continue;            /* an end node */
        nle = le; nrs = rs;
        for (j = L->src_first[i]; j < L->src_first[i + 1]; j++) {       /* collect_contain, overlap_rate 0.1 */
            const lhit *h = &L->h[j];
            if (h->del) continue;
            if (h->qs != 0 && h->qe != rl) {
                if (h->qs < le && h->qe > le && le - h->qs > (0.1f * (h->qe - h->qs))) { if (h->qe > le && h->qe > nle) nle = h->qe; }
                if (h->qs < rs && h->qe > rs && h->qe - rs > (0.1f * (h->qe - h->qs))) { if (h->qs < rs && h->qs < nrs) nrs = h->qs; }
            }
        }
        le = nle; rs = nrs;
        if (le > rs && (le - rs >= rl * shift_rate)) continue;   /* a normal read */
        if (le <= rs) { delete_all_edges(L, i); continue; }      /* simple chimeric read */
        /* complex case: not restated (see the header) */
    }
}

static void hit_cut(lay_t *L)
{
    int i;
    for (i = 0; i < L->n_h; i++) {
        lhit *p = &L->h[i];
        if (p->del || L->rdel[p->qn] || L->rdel[p->tn]) continue;
        if (!(p->qe - p->qs >= MIN_OVLP && p->te - p->ts >= MIN_OVLP)) p->del = 1;
    }
}

static void hit_flt(lay_t *L)
{
    int i, j;
    larc t;
    for (i = 0; i < L->n; i++) {
        int kept = 0;
        for (j = L->src_first[i]; j < L->src_first[i + 1]; j++) {
            lhit *h = &L->h[j];
            int r;
            if (h->del) continue;
            if (L->rdel[h->qn] || L->rdel[h->tn]) continue;
            r = hit2arc(h, L->len[h->qn], L->len[h->tn], &t);
            if (r >= 0 || r == HT_QCONT || r == HT_TCONT) { h->del = 0; kept++; }
            else { h->del = 1; delete_single_edge(L, h->tn, h->qn); }
        }
        if (kept == 0) L->rdel[i] = 1;
    }
}

static void hit_contained(lay_t *L)
{
    int i, j;
    larc t;
    for (i = 0; i < L->n; i++) {
        if (L->rdel[i]) continue;
        for (j = L->src_first[i]; j < L->src_first[i + 1]; j++) {
            lhit *h = &L->h[j];
            int r;
            if (L->rdel[h->qn] || L->rdel[h->tn]) continue;
            if (h->del) continue;
            r = hit2arc(h, L->len[h->qn], L->len[h->tn], &t);
            if (r == HT_QCONT) { h->del = 1; delete_single_edge(L, h->tn, h->qn); delete_all_edges(L, h->qn); }
            else if (r == HT_TCONT) { h->del = 1; delete_single_edge(L, h->tn, h->qn); delete_all_edges(L, h->tn); }
        }
    }
    for (i = 0; i < L->n; i++) {
        int m = 0;
        for (j = L->src_first[i]; j < L->src_first[i + 1]; j++) {
            lhit *h = &L->h[j];
            if (h->del) continue;
            if (!L->rdel[h->qn] && !L->rdel[h->tn]) { h->del = 0; m++; } else h->del = 1;
        }
        if (m == 0) L->rdel[i] = 1;
    }
}

/* ---- the string graph */
static int arc_cmp(const void *a, const void *b)
{
    const larc *x = (const larc *)a, *y = (const larc *)b;
    if (x->u != y->u) return x->u < y->u ? -1 : 1;
    if (x->len != y->len) return x->len < y->len ? -1 : 1;
    return x->v < y->v ? -1 : (x->v > y->v);
}
static void g_cleanup(lay_t *L)   /* asg_cleanup: drop deleted arcs and arcs of deleted reads, sort, index */
{
    int e, m = 0, v;
    for (e = 0; e < L->n_arc; e++) {
        const larc *a = &L->arc[e];
        if (!a->del && !L->sdel[a->u >> 1] && !L->sdel[a->v >> 1]) L->arc[m++] = *a;
    }
    L->n_arc = m;
    qsort(L->arc, (size_t)m, sizeof(larc), arc_cmp);
    for (v = 0, e = 0; v < 2 * L->n; v++) { L->a_first[v] = e; while (e < m && (int)L->arc[e].u == v) e++; }
    L->a_first[2 * L->n] = m;
}
#define ARC_N(L, v) ((L)->a_first[(v) + 1] - (L)->a_first[(v)])
#define ARC_A(L, v) ((L)->arc + (L)->a_first[(v)])
static void g_symm(lay_t *L)
{
    int v, i, e, changed = 0;
    int *cnt = (int *)calloc((size_t)2 * L->n, sizeof(int));
    for (v = 0; v < 2 * L->n; v++) {          /* asg_arc_del_multi */
        larc *av = ARC_A(L, v); int nv = ARC_N(L, v);
        if (nv < 2) continue;
        for (i = nv - 1; i >= 0; --i) ++cnt[av[i].v];
        for (i = nv - 1; i >= 0; --i) if (--cnt[av[i].v] != 0) { av[i].del = 1; changed = 1; }
    }
    free(cnt);
    if (changed) g_cleanup(L);
    changed = 0;
    for (e = 0; e < L->n_arc; e++) {          /* asg_arc_del_asymm */
        const int v = (int)(L->arc[e].v ^ 1), u = (int)(L->arc[e].u ^ 1);
        larc *av = ARC_A(L, v); int nv = ARC_N(L, v);
        for (i = 0; i < nv; i++) if ((int)av[i].v == u) break;
        if (i == nv) { L->arc[e].del = 1; changed = 1; }
    }
    if (changed) g_cleanup(L);
}
static void arc_del_pair(lay_t *L, uint32_t v, uint32_t w)   /* asg_arc_del(g, v, w, 1) */
{
    larc *av = ARC_A(L, v); int nv = ARC_N(L, v), i;
    for (i = 0; i < nv; i++) if (av[i].v == w) av[i].del = 1;
}
static void seq_del(lay_t *L, uint32_t s)                    /* asg_seq_del */
{
    int k, i;
    L->sdel[s] = 1;
    for (k = 0; k < 2; k++) {
        const uint32_t v = s << 1 | (uint32_t)k;
        larc *av = ARC_A(L, v); int nv = ARC_N(L, v);
        for (i = 0; i < nv; i++) { av[i].del = 1; arc_del_pair(L, av[i].v ^ 1, v ^ 1); }
    }
}

static void del_trans(lay_t *L)
{
    uint8_t *mark = (uint8_t *)calloc((size_t)2 * L->n, 1);
    int v, i, j, n_red = 0;
    for (v = 0; v < 2 * L->n; v++) {
        larc *av = ARC_A(L, v); int nv = ARC_N(L, v), Lmax;
        if (nv == 0) continue;
        if (L->sdel[v >> 1]) { for (i = 0; i < nv; i++) { av[i].del = 1; n_red++; } continue; }
        for (i = 0; i < nv; i++) mark[av[i].v] = 1;
        Lmax = av[nv - 1].len + GAP_FUZZ;
        for (i = 0; i < nv; i++) {
            const uint32_t w = av[i].v;
            larc *aw = ARC_A(L, w); int nw = ARC_N(L, w);
            if (mark[av[i].v] != 1) continue;
            for (j = 0; j < nw && aw[j].len + av[i].len <= Lmax; j++) if (mark[aw[j].v]) mark[aw[j].v] = 2;
        }
        for (i = 0; i < nv; i++) { if (mark[av[i].v] == 2) { av[i].del = 1; n_red++; } mark[av[i].v] = 0; }
    }
    free(mark);
    if (n_red) { g_cleanup(L); g_symm(L); }
}

enum { ET_MERGEABLE = 0, ET_TIP = 1, ET_MULTI_OUT = 2, ET_MULTI_NEI = 3 };
static int is_utg_end(const lay_t *L, uint32_t v, uint32_t *lw)
{
    const larc *av = ARC_A(L, v ^ 1), *aw; int nv0 = ARC_N(L, v ^ 1), i, i0 = -1, nv = 0, nw0, nw = 0;
    uint32_t w;
    for (i = 0; i < nv0; i++) if (!av[i].del) { i0 = i; nv++; }
    if (nv == 0) return ET_TIP;
    if (nv > 1) return ET_MULTI_OUT;
    if (lw) *lw = av[i0].v;
    w = av[i0].v ^ 1;
    nw0 = ARC_N(L, w); aw = ARC_A(L, w);
    for (i = 0; i < nw0; i++) if (!aw[i].del) nw++;
    if (nw != 1) return ET_MULTI_NEI;
    return ET_MERGEABLE;
}
static void cut_tip(lay_t *L)
{
    int v, i, cnt = 0;
    uint32_t a[MAX_SHORT_TIP + 2];
    for (v = 0; v < 2 * L->n; v++) {
        int na = 0, max_ext = MAX_SHORT_TIP, ret;
        uint32_t x = (uint32_t)v, lw = 0;
        if (L->sdel[v >> 1]) continue;
        if (is_utg_end(L, (uint32_t)v, NULL) != ET_TIP) continue;
        a[na++] = x;
        do {                                  /* asg_extend */
            ret = is_utg_end(L, x ^ 1, &lw);
            if (ret != 0) break;
            a[na++] = lw;
            x = lw;
        } while (--max_ext > 0);
        if (ret == ET_MERGEABLE) continue;    /* not a short unitig */
        for (i = 0; i < na; i++) seq_del(L, a[i] >> 1);
        cnt++;
    }
    if (cnt) g_cleanup(L);
}

/* ---- unitigs and their polishing.  An element is vertex << 32 | length (the node's share of the unitig); ~0 = skipped */
typedef struct { uint64_t *a; int n; int circ; } utg_t;

static int get_edge_from_source(const lay_t *L, uint32_t query, uint32_t target, larc *t)
{
    int i; const int qn = (int)(query >> 1);
    for (i = L->src_first[qn]; i < L->src_first[qn + 1]; i++) {       /* (deleted hits too, as the reference) */
        const lhit *h = &L->h[i];
        const int r = hit2arc(h, L->len[h->qn], L->len[h->tn], t);
        if (r < 0) continue;
        if (t->u != query || t->v != target) continue;
        return 1;
    }
    return 0;
}
static int get_specific_edge(const lay_t *L, int use_graph, uint32_t query, uint32_t target, larc *t)
{
    int k;
    if (use_graph) {
        const larc *av = ARC_A(L, query); int nv = ARC_N(L, query);
        for (k = 0; k < nv; k++) { if (av[k].del) continue; if (av[k].v == target) { *t = av[k]; return 1; } }
    }
    return get_edge_from_source(L, query, target, t);
}
static void get_overlap_len(const lay_t *L, int r, uint32_t *exact_len, uint32_t *inexact_len)
{
    int i;
    *exact_len = *inexact_len = 0;
    for (i = L->src_first[r]; i < L->src_first[r + 1]; i++) {
        const uint32_t l = (uint32_t)(L->h[i].qe - L->h[i].qs);
        if (L->h[i].el == 1) *exact_len += l; else *inexact_len += l;
    }
}
static void reduce_utg(const lay_t *L, utg_t *U)
{
    int i, m = 0;
    for (i = 0; i < U->n; i++) if (U->a[i] != ~0ull) U->a[m++] = U->a[i];
    U->n = m;
    for (i = 0; i + 1 < U->n; i++) {
        const uint32_t v = (uint32_t)(U->a[i] >> 32), w = (uint32_t)(U->a[i + 1] >> 32);
        larc t; uint32_t l = 0;
        if (get_specific_edge(L, 1, v, w, &t)) l = (uint32_t)t.len;
        U->a[i] = (uint64_t)v << 32 | l;
    }
    if (U->n > 0) { const uint32_t v = (uint32_t)(U->a[U->n - 1] >> 32); U->a[U->n - 1] = (uint64_t)v << 32 | (uint32_t)L->len[v >> 1]; }
}

static void polish_unitig(const lay_t *L, utg_t *U)
{
    uint32_t pre; int i, k, pre_i = 0, skip = 0;
    if (U->n < 3) return;
    pre = (uint32_t)(U->a[0] >> 32);
    for (i = 1; i < U->n - 1; i++) {
        uint32_t v, afte, exact_len, inexact_len, min_inexact = 0xffffffffu, max_exact = 0;
        larc pE, aE, t;
        int afte_i = -1, idx;
        if (U->a[i] == ~0ull) continue;
        v = (uint32_t)(U->a[i] >> 32); afte = (uint32_t)(U->a[i + 1] >> 32);
        memset(&pE, 0, sizeof(pE)); memset(&aE, 0, sizeof(aE));
        get_specific_edge(L, pre_i == i - 1, v ^ 1, pre ^ 1, &pE);
        get_specific_edge(L, 1, v, afte, &aE);
        if (pE.el == 1 && aE.el == 1) { pre = v; pre_i = i; continue; }
        /* pre is a good read: the next read that overlaps it exactly (detect_exact_ovec) */
        for (idx = i + 1; idx < U->n; idx++) {
            const uint32_t dest = (uint32_t)(U->a[idx] >> 32);
            if (!get_edge_from_source(L, pre, dest, &t)) break;
            if (t.el != 1) continue;
            afte_i = idx;
            break;
        }
        if (afte_i < 0) { pre = v; pre_i = i; continue; }
        afte = (uint32_t)(U->a[afte_i] >> 32);
        for (k = i; k < afte_i; k++) {
            get_overlap_len(L, (int)(U->a[k] >> 33), &exact_len, &inexact_len);
            if (inexact_len < min_inexact) { min_inexact = inexact_len; max_exact = exact_len; }
        }
        get_overlap_len(L, (int)(pre >> 1), &exact_len, &inexact_len);
        if (inexact_len > min_inexact || (inexact_len == min_inexact && exact_len <= max_exact)) { pre = v; pre_i = i; continue; }
        get_overlap_len(L, (int)(afte >> 1), &exact_len, &inexact_len);
        if (inexact_len > min_inexact || (inexact_len == min_inexact && exact_len <= max_exact)) { pre = v; pre_i = i; continue; }
        for (k = i; k < afte_i; k++) { U->a[k] = ~0ull; skip++; }
    }
    if (skip) reduce_utg(L, U);
}

static char base_of(const lay_t *L, uint32_t v, int p)   /* base p of read v>>1 on strand v&1 */
{
    const char *s = L->seq[v >> 1]; const int n = L->len[v >> 1];
    char c;
    if (!(v & 1)) return s[p];
    c = s[n - 1 - p];
    return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
}
/* how many of the unitig's earlier reads that cover the stretch of read cur in front of read next agree with it base for base;
 * 1 counted, 0 an element is skipped, -1 no edge from cur to next */
static int consensus_rate(const lay_t *L, const utg_t *U, int cur_i, int next_i, int *r_match, int *r_total)
{
    larc t, e; uint32_t v, w; int v_beg, v_end, j, p;
    *r_match = *r_total = 0;
    if (cur_i < 1) return -1;
    if (U->a[cur_i] == ~0ull || U->a[next_i] == ~0ull) return 0;
    v = (uint32_t)(U->a[cur_i] >> 32); w = (uint32_t)(U->a[next_i] >> 32);
    if (!get_edge_from_source(L, v, w, &t)) return -1;
    v_beg = 0; v_end = t.len - 1;
    for (j = cur_i - 1; j >= 0; j--) {
        int w_beg, w_end;
        uint32_t wj;
        if (U->a[j] == ~0ull) continue;
        wj = (uint32_t)(U->a[j] >> 32);
        if (!get_edge_from_source(L, wj, v, &e)) break;
        w_beg = e.len;
        w_end = w_beg + v_end < L->len[wj >> 1] - 1 ? w_beg + v_end : L->len[wj >> 1] - 1;
        if (w_end - w_beg != v_end - v_beg) break;     /* does not cover the whole stretch */
        (*r_total)++;
        if (e.el == 1) (*r_match)++;
        else {
            int same = 1;
            for (p = 0; p <= v_end - v_beg && same; p++) if (base_of(L, v, v_beg + p) != base_of(L, wj, w_beg + p)) same = 0;
            if (same) (*r_match)++;
        }
    }
    return 1;
}
static void polish_unitig_advance(const lay_t *L, utg_t *U)
{
    int i, k, skip = 0, match_v, total_v;
    if (U->n < 3) return;
    for (i = 1; i < U->n - 1; i++) {
        int max_i, match_max; double match_rate, match_rate_max;
        if (consensus_rate(L, U, i, i + 1, &match_v, &total_v) != 1) continue;
        match_rate = total_v == 0 ? 0 : (double)match_v / (double)total_v;
        if (match_v >= total_v * 0.5 && total_v > 0 && match_v > 0) continue;     /* most reads support this one */
        max_i = i; match_max = match_v; match_rate_max = match_rate;
        for (k = i - 1; k >= 0; k--) {
            if (U->a[k] == ~0ull) continue;
            if (consensus_rate(L, U, k, i + 1, &match_v, &total_v) < 0) break;
            if (total_v == 0) break;
            match_rate = (double)match_v / (double)total_v;
            if (match_rate > match_rate_max || (match_rate == match_rate_max && match_v > match_max)) { max_i = k; match_max = match_v; match_rate_max = match_rate; }
        }
        for (k = max_i + 1; k <= i; k++) { if (U->a[k] == ~0ull) continue; U->a[k] = ~0ull; skip++; }
    }
    if (skip) reduce_utg(L, U);
}

/* hits: the final overlaps, ordered pairs (both directions present), inclusive coordinates, y in strand coordinates.
 * -> pieces (read, strand, bases taken) of every contig; returns the number of contigs */
int orc_layout_graph(const char *const *seq, const int *len, int n, const orc_ovl *hit, int n_hit, int min_reads, int32_t *piece_read,
                     uint8_t *piece_rev, int32_t *piece_len, int32_t *contig_first, int piece_cap, int contig_cap)
{
    lay_t L;
    int i, v, n_piece = 0, n_contig = 0;
    int *cnt = (int *)calloc((size_t)n + 1, sizeof(int)), *fill;
    uint8_t *mark;
    memset(&L, 0, sizeof(L));
    L.n = n; L.len = len; L.seq = seq;
    L.h = (lhit *)malloc(sizeof(lhit) * (size_t)(n_hit + 1)); L.n_h = n_hit;
    L.src_first = (int *)calloc((size_t)n + 2, sizeof(int));
    L.rdel = (uint8_t *)calloc((size_t)n + 1, 1); L.sdel = (uint8_t *)calloc((size_t)n + 1, 1);
    for (i = 0; i < n_hit; i++) cnt[hit[i].q]++;
    for (i = 0; i < n; i++) L.src_first[i + 1] = L.src_first[i] + cnt[i];
    fill = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    memcpy(fill, L.src_first, sizeof(int) * (size_t)(n + 1));
    for (i = 0; i < n_hit; i++) {
        const orc_ovl *o = &hit[i];
        lhit *h = &L.h[fill[o->q]++];
        const int tl = len[o->t];
        h->qn = (int32_t)o->q; h->tn = (int32_t)o->t; h->qs = o->x_s; h->qe = o->x_e + 1; h->rev = o->rev; h->el = o->exact; h->del = 0; h->pad = 0;
        if (o->rev) { h->ts = tl - o->y_e - 1; h->te = tl - o->y_s; } else { h->ts = o->y_s; h->te = o->y_e + 1; }
    }
    free(fill); free(cnt);
    for (i = 0; i < n; i++) {       /* sources[i] sorted by target id (overlap_region_sort_y_id before push_final_overlaps) */
        int a = L.src_first[i], b = L.src_first[i + 1], x, y;
        for (x = a + 1; x < b; x++) { lhit t = L.h[x]; for (y = x; y > a && L.h[y - 1].tn > t.tn; y--) L.h[y] = L.h[y - 1]; L.h[y] = t; }
    }
    normalize(&L);
    chimeric(&L);
    hit_cut(&L);
    hit_flt(&L);
    hit_contained(&L);
    /* ma_sg_gen */
    L.arc = (larc *)malloc(sizeof(larc) * (size_t)(n_hit + 1));
    L.a_first = (int *)calloc((size_t)2 * n + 2, sizeof(int));
    for (i = 0; i < n; i++) L.sdel[i] = L.rdel[i];
    for (i = 0; i < n_hit; i++) {
        const lhit *h = &L.h[i];
        larc t;
        if (h->del) continue;
        if (hit2arc(h, len[h->qn], len[h->tn], &t) >= 0) L.arc[L.n_arc++] = t;
    }
    g_cleanup(&L);
    del_trans(&L);
    cut_tip(&L);
    if (getenv("ORC_DEBUG_LAYOUT")) {
        for (i = 0; i < L.n_arc; i++) fprintf(stderr, "ARC %u%c->%u%c ol=%d el=%u len=%d\n", L.arc[i].u >> 1, "+-"[L.arc[i].u & 1], L.arc[i].v >> 1, "+-"[L.arc[i].v & 1], L.arc[i].ol, L.arc[i].el, L.arc[i].len);
    }
    /* ma_ug_gen: vertices in increasing order; the unitig through the first unvisited one, from its start */
    mark = (uint8_t *)calloc((size_t)2 * n + 1, 1);
    for (v = 0; v < 2 * n; v++) {
        utg_t U; int cap = n + 2, nq = 0, head, k;
        uint64_t *q;
        uint32_t w, x, start, end;
        if (L.sdel[v >> 1] || mark[v]) continue;
        if (ARC_N(&L, v) == 0 && ARC_N(&L, v ^ 1) != 0) continue;
        q = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(2 * cap + 2));
        head = cap;                       /* a deque: forward pushes go right of head, backward pushes left */
        mark[v] = 1;
        start = (uint32_t)v; end = (uint32_t)v ^ 1;
        w = (uint32_t)v;
        while (1) {
            if (ARC_N(&L, w) != 1) break;
            x = ARC_A(&L, w)[0].v;
            if (ARC_N(&L, x ^ 1) != 1) break;
            mark[x] = mark[w ^ 1] = 1;
            q[head + nq++] = (uint64_t)w << 32 | (uint32_t)ARC_A(&L, w)[0].len;
            end = x ^ 1;
            w = x;
            if (x == (uint32_t)v) break;
        }
        U.circ = 0;
        if (start != (end ^ 1) || nq == 0) {
            q[head + nq++] = (uint64_t)(end ^ 1) << 32 | (uint32_t)len[end >> 1];
            x = (uint32_t)v;
            while (1) {
                if (ARC_N(&L, x ^ 1) != 1) break;
                w = ARC_A(&L, x ^ 1)[0].v ^ 1;
                if (ARC_N(&L, w) != 1) break;
                mark[x] = mark[w ^ 1] = 1;
                head--; nq++;
                q[head] = (uint64_t)w << 32 | (uint32_t)ARC_A(&L, w)[0].len;
                start = w;
                x = w;
                if (nq > 2 * n) break;
            }
            mark[start] = mark[end] = 1;
        } else U.circ = 1;
        U.a = q + head; U.n = nq;
        if (nq >= min_reads && n_contig < contig_cap && n_piece + nq <= piece_cap) {
            if (!U.circ) { polish_unitig(&L, &U); polish_unitig_advance(&L, &U); }
            contig_first[n_contig++] = n_piece;
            for (k = 0; k < U.n; k++) {
                const uint32_t vv = (uint32_t)(U.a[k] >> 32);
                piece_read[n_piece] = (int32_t)(vv >> 1); piece_rev[n_piece] = (uint8_t)(vv & 1); piece_len[n_piece] = (int32_t)(uint32_t)U.a[k];
                n_piece++;
            }
        }
        free(q);
    }
    contig_first[n_contig] = n_piece;
    free(mark); free(L.h); free(L.src_first); free(L.rdel); free(L.sdel); free(L.arc); free(L.a_first);
    return n_contig;
}
