// layout.h -- host side of the assembly's last step: final overlaps of one read set -> the reads (strand, prefix length) each contig
// is spelled from.  hifiasm-0.14's own order of business for a set of tens to hundreds of reads (Overlaps.cpp; restated for the tests
// in oracle/layout.c, which this must match contig for contig):
//   detect_chimeric_reads :1698   a read whose overlaps from its left end and from its right end do not reach each other is dropped
//   ma_hit_cut :1785, ma_hit_flt :1132   overlaps below 50 bases, internal matches
//   ma_hit_contained_advance :1031       containment in read order (a read already removed contains nobody)
//   ma_sg_gen :2152 / ma_hit2arc Overlaps.h:178, asg_arc_del_trans :4531 (fuzz 1000) + asg_symm :342, asg_cut_tip :4666 (3 reads)
//   ma_ug_gen :7759                      unitigs in vertex order
//   polish_unitig :8480, polish_unitig_advance :8893   a read joined by an inexact overlap is skipped when reads around it overlap
//                                        exactly / when the unitig's earlier reads disagree with it base for base
// None of clean_graph's other steps (:27087-27350) changes the graph of such a set (traced with the reference's code), so they are
// not here.  The graph is a few hundred arcs: plain vectors, one set per host thread.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace fsv_layout {

struct Hit { int32_t qn, tn, qs, qe, ts, te; uint8_t rev, el, del; };       // qe / te exclusive; target in its own forward coordinates
struct Arc { uint32_t u, v; int32_t len, ol; uint8_t el, del; };             // u -> v; len = the node's share, ol = overlap length
struct PieceOut { uint32_t read, rev, len; };

// bases of the set's corrected reads, 2 bits each (the device store's layout), fetched only when an inexact overlap needs comparing
struct ReadBases {
    const uint32_t *words = nullptr;      // host copy of the store (may be null when every overlap is exact)
    const uint32_t *word_off = nullptr;   // per read of the set
    const int32_t *len = nullptr;
    uint32_t base(uint32_t v, int p) const    // base p of read v >> 1 on strand v & 1
    {
        const uint32_t r = v >> 1;
        const int q = (v & 1) ? len[r] - 1 - p : p;
        const uint32_t b = (words[word_off[r] + ((uint32_t)q >> 4)] >> (((uint32_t)q & 15u) << 1)) & 3u;
        return (v & 1) ? 3u - b : b;
    }
};

enum { HT_INT = -1, HT_QCONT = -2, HT_TCONT = -3, HT_SHORT = -4 };
constexpr int MAX_HANG = 1000, MIN_OVLP = 50, GAP_FUZZ = 1000, MAX_SHORT_TIP = 3;
constexpr float INT_FRAC = 0.8f;

class Graph {
public:
    Graph(const int32_t *len, int n, const ReadBases &rb) : len_(len), n_(n), rb_(rb), src_first_(n + 1, 0), rdel_(n, 0), sdel_(n, 0), a_first_(2 * n + 1, 0) {}

    int hit2arc(const Hit &h, Arc &p) const
    {
        const int ql = len_[h.qn], tl = len_[h.tn], qs = h.qs;
        int tl5, tl3;
        if (h.rev) { tl5 = tl - h.te; tl3 = h.ts; } else { tl5 = h.ts; tl3 = tl - h.te; }
        const int ext5 = std::min(qs, tl5), ext3 = std::min(ql - h.qe, tl3);
        if (ext5 > MAX_HANG || ext3 > MAX_HANG || h.qe - qs < (h.qe - qs + ext5 + ext3) * INT_FRAC || h.te - h.ts < (h.te - h.ts + ext5 + ext3) * INT_FRAC) return HT_INT;
        uint32_t u, v; int l;
        if (qs <= tl5 && ql - h.qe <= tl3) return HT_QCONT;
        else if (qs >= tl5 && ql - h.qe >= tl3) return HT_TCONT;
        else if (qs > tl5) { u = 0; v = h.rev ? 1 : 0; l = qs - tl5; }
        else { u = 1; v = h.rev ? 0 : 1; l = (ql - h.qe) - tl3; }
        if (h.qe - qs + ext5 + ext3 < MIN_OVLP || h.te - h.ts + ext5 + ext3 < MIN_OVLP) return HT_SHORT;
        p.u = u | (uint32_t)h.qn << 1; p.v = v | (uint32_t)h.tn << 1; p.len = l; p.ol = ql - l; p.el = h.el; p.del = 0;
        return l;
    }

    // hits grouped by query, each group sorted by target (the order hifiasm pushes them in)
    void set_hits(std::vector<Hit> &&h)
    {
        // counting sort by query, then each query's few dozen hits by target (a pair has one hit: no ties)
        for (const Hit &x : h) src_first_[x.qn + 1]++;
        for (int i = 0; i < n_; i++) src_first_[i + 1] += src_first_[i];
        h_.resize(h.size());
        std::vector<int> at(src_first_.begin(), src_first_.end() - 1);
        for (const Hit &x : h) h_[at[x.qn]++] = x;
        for (int q = 0; q < n_; q++)
            for (int i = src_first_[q] + 1; i < src_first_[q + 1]; i++) {
                const Hit t = h_[i];
                int j = i;
                for (; j > src_first_[q] && h_[j - 1].tn > t.tn; j--) h_[j] = h_[j - 1];
                h_[j] = t;
            }
    }

    void build()
    {
        normalize(); chimeric(); hit_cut(); hit_flt(); hit_contained();
        sdel_ = rdel_;
        for (const Hit &h : h_) { Arc t; if (!h.del && hit2arc(h, t) >= 0) arc_.push_back(t); }
        cleanup();
        del_trans();
        cut_tip();
    }

    // unitigs of at least min_reads reads, polished -> pieces per contig
    void unitigs(int min_reads, std::vector<std::vector<PieceOut>> &out)
    {
        std::vector<uint8_t> mark(2 * n_, 0);
        for (int v = 0; v < 2 * n_; v++) {
            if (sdel_[v >> 1] || mark[v]) continue;
            if (arc_n(v) == 0 && arc_n(v ^ 1) != 0) continue;
            std::vector<uint64_t> fwd, back;
            mark[v] = 1;
            uint32_t start = (uint32_t)v, end = (uint32_t)v ^ 1, w = (uint32_t)v, x;
            while (true) {
                if (arc_n(w) != 1) break;
                x = arc_a(w)[0].v;
                if (arc_n(x ^ 1) != 1) break;
                mark[x] = mark[w ^ 1] = 1;
                fwd.push_back((uint64_t)w << 32 | (uint32_t)arc_a(w)[0].len);
                end = x ^ 1;
                w = x;
                if (x == (uint32_t)v) break;
            }
            const bool circ = !(start != (end ^ 1) || fwd.empty());
            if (!circ) {
                fwd.push_back((uint64_t)(end ^ 1) << 32 | (uint32_t)len_[end >> 1]);
                x = (uint32_t)v;
                while ((int)back.size() <= 2 * n_) {
                    if (arc_n(x ^ 1) != 1) break;
                    w = arc_a(x ^ 1)[0].v ^ 1;
                    if (arc_n(w) != 1) break;
                    mark[x] = mark[w ^ 1] = 1;
                    back.push_back((uint64_t)w << 32 | (uint32_t)arc_a(w)[0].len);
                    start = w;
                    x = w;
                }
                mark[start] = mark[end] = 1;
            }
            std::vector<uint64_t> u(back.rbegin(), back.rend());
            u.insert(u.end(), fwd.begin(), fwd.end());
            if ((int)u.size() < min_reads) continue;
            if (!circ) { polish(u); polish_advance(u); }
            std::vector<PieceOut> pcs;
            for (uint64_t e : u) pcs.push_back(PieceOut{(uint32_t)(e >> 33), (uint32_t)(e >> 32) & 1u, (uint32_t)e});
            out.push_back(std::move(pcs));
        }
    }

    bool any_inexact() const { for (const Hit &h : h_) if (!h.el) return true; return false; }

private:
    const int32_t *len_; int n_; const ReadBases &rb_;
    std::vector<Hit> h_;
    std::vector<int> src_first_;
    std::vector<uint8_t> rdel_, sdel_;
    std::vector<Arc> arc_;
    std::vector<int> a_first_;
    static constexpr uint64_t SKIP = ~0ull;

    int arc_n(uint32_t v) const { return a_first_[v + 1] - a_first_[v]; }
    Arc *arc_a(uint32_t v) { return arc_.data() + a_first_[v]; }
    const Arc *arc_a(uint32_t v) const { return arc_.data() + a_first_[v]; }

    Hit *find_hit(int qn, int tn)      // a query's hits are sorted by target
    {
        int lo = src_first_[qn], hi = src_first_[qn + 1];
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (h_[mid].tn < tn) lo = mid + 1; else hi = mid; }
        return lo < src_first_[qn + 1] && h_[lo].tn == tn ? &h_[lo] : nullptr;
    }
    void delete_single_edge(int qn, int tn) { if (Hit *t = find_hit(qn, tn)) t->del = 1; }
    void delete_all_edges(int qn)
    {
        for (int i = src_first_[qn]; i < src_first_[qn + 1]; i++) { h_[i].del = 1; delete_single_edge(h_[i].tn, qn); }
        rdel_[qn] = 1;
    }

    // normalize_ma_hit_t_single_side_advance (Overlaps.cpp:450-515): the two directions of a pair are made one overlap -- the direction
    // with the longer query interval stands and the other becomes its mirror image (equal lengths: the lower read's stands); a hit
    // without a partner is deleted.  (The final pass's gapped re-chain chains either direction from its own side.)
    void normalize()
    {
        for (int i = 0; i < n_; i++)
            for (int j = src_first_[i]; j < src_first_[i + 1]; j++) {
                Hit &h = h_[j];
                if (Hit *r = find_hit(h.tn, h.qn)) {
                    const bool is_del = h.del || r->del;
                    const int q0 = h.qe - h.qs, q1 = r->qe - r->qs;
                    if ((q0 == q1 && h.qn < h.tn) || q0 > q1) { r->qs = h.ts; r->qe = h.te; r->ts = h.qs; r->te = h.qe; r->rev = h.rev; r->el = h.el; }
                    h.del = r->del = is_del ? 1 : 0;
                } else h.del = 1;
            }
    }
    void chimeric()
    {
        const float shift_rate = (float)(0.001f * 2.0);
        for (int i = 0; i < n_; i++) {
            const int64_t rl = len_[i];
            int64_t ls = rl, le = 0, rs = rl, re = 0;
            for (int j = src_first_[i]; j < src_first_[i + 1]; j++) {
                const Hit &h = h_[j];
                if (h.del) continue;
                if (h.qs == 0) { ls = std::min<int64_t>(ls, h.qs); le = std::max<int64_t>(le, h.qe); }
                if (h.qe == rl) { rs = std::min<int64_t>(rs, h.qs); re = std::max<int64_t>(re, h.qe); }
            }
            if (ls == rl || rs == rl) continue;            // an end node
            int64_t nle = le, nrs = rs;
            for (int j = src_first_[i]; j < src_first_[i + 1]; j++) {       // collect_contain, overlap_rate 0.1
                const Hit &h = h_[j];
                if (h.del || h.qs == 0 || h.qe == rl) continue;
                if (h.qs < le && h.qe > le && le - h.qs > (0.1f * (h.qe - h.qs)) && h.qe > nle) nle = h.qe;
                if (h.qs < rs && h.qe > rs && h.qe - rs > (0.1f * (h.qe - h.qs)) && h.qs < nrs) nrs = h.qs;
            }
            le = nle; rs = nrs;
            if (le > rs && (le - rs >= rl * shift_rate)) continue;   // a normal read
            if (le <= rs) delete_all_edges(i);                       // a simple chimeric read
            // (the complex case -- the two sides meet in under 0.2 % of the read and some spanning overlap fails a window check there --
            // is taken as "not chimeric": final overlaps are exact or were verified window by window in the last correction round)
        }
    }
    void hit_cut()
    {
        for (Hit &p : h_) {
            if (p.del || rdel_[p.qn] || rdel_[p.tn]) continue;
            if (!(p.qe - p.qs >= MIN_OVLP && p.te - p.ts >= MIN_OVLP)) p.del = 1;
        }
    }
    void hit_flt()
    {
        Arc t;
        for (int i = 0; i < n_; i++) {
            int kept = 0;
            for (int j = src_first_[i]; j < src_first_[i + 1]; j++) {
                Hit &h = h_[j];
                if (h.del || rdel_[h.qn] || rdel_[h.tn]) continue;
                const int r = hit2arc(h, t);
                if (r >= 0 || r == HT_QCONT || r == HT_TCONT) kept++;
                else { h.del = 1; delete_single_edge(h.tn, h.qn); }
            }
            if (kept == 0) rdel_[i] = 1;
        }
    }
    void hit_contained()
    {
        Arc t;
        for (int i = 0; i < n_; i++) {
            if (rdel_[i]) continue;
            for (int j = src_first_[i]; j < src_first_[i + 1]; j++) {
                Hit &h = h_[j];
                if (rdel_[h.qn] || rdel_[h.tn] || h.del) continue;
                const int r = hit2arc(h, t);
                if (r == HT_QCONT) { h.del = 1; delete_single_edge(h.tn, h.qn); delete_all_edges(h.qn); }
                else if (r == HT_TCONT) { h.del = 1; delete_single_edge(h.tn, h.qn); delete_all_edges(h.tn); }
            }
        }
        for (int i = 0; i < n_; i++) {
            int m = 0;
            for (int j = src_first_[i]; j < src_first_[i + 1]; j++) {
                Hit &h = h_[j];
                if (h.del) continue;
                if (!rdel_[h.qn] && !rdel_[h.tn]) m++; else h.del = 1;
            }
            if (m == 0) rdel_[i] = 1;
        }
    }

    void cleanup()      // asg_cleanup: deleted arcs and arcs of deleted reads go, the rest is sorted by (vertex, length) and indexed
    {
        size_t m = 0;
        for (const Arc &a : arc_) if (!a.del && !sdel_[a.u >> 1] && !sdel_[a.v >> 1]) arc_[m++] = a;
        arc_.resize(m);
        std::sort(arc_.begin(), arc_.end(), [](const Arc &x, const Arc &y) { return x.u != y.u ? x.u < y.u : x.len != y.len ? x.len < y.len : x.v < y.v; });
        size_t e = 0;
        for (int v = 0; v < 2 * n_; v++) { a_first_[v] = (int)e; while (e < m && (int)arc_[e].u == v) e++; }
        a_first_[2 * n_] = (int)m;
    }
    void symm()
    {
        bool changed = false;
        std::vector<int> cnt(2 * n_, 0);
        for (int v = 0; v < 2 * n_; v++) {          // asg_arc_del_multi: of several arcs to one vertex the shortest node length stays
            Arc *av = arc_a(v); const int nv = arc_n(v);
            if (nv < 2) continue;
            for (int i = nv - 1; i >= 0; --i) ++cnt[av[i].v];
            for (int i = nv - 1; i >= 0; --i) if (--cnt[av[i].v] != 0) { av[i].del = 1; changed = true; }
        }
        if (changed) cleanup();
        changed = false;
        for (Arc &a : arc_) {                        // asg_arc_del_asymm
            const uint32_t v = a.v ^ 1, u = a.u ^ 1;
            const Arc *av = arc_a(v); const int nv = arc_n(v);
            int i = 0;
            for (; i < nv; i++) if (av[i].v == u) break;
            if (i == nv) { a.del = 1; changed = true; }
        }
        if (changed) cleanup();
    }
    void seq_del(uint32_t s)
    {
        sdel_[s] = 1;
        for (uint32_t k = 0; k < 2; k++) {
            const uint32_t v = s << 1 | k;
            Arc *av = arc_a(v); const int nv = arc_n(v);
            for (int i = 0; i < nv; i++) {
                av[i].del = 1;
                Arc *aw = arc_a(av[i].v ^ 1); const int nw = arc_n(av[i].v ^ 1);
                for (int j = 0; j < nw; j++) if (aw[j].v == (v ^ 1)) aw[j].del = 1;
            }
        }
    }
    void del_trans()
    {
        std::vector<uint8_t> mark(2 * n_, 0);
        int n_red = 0;
        for (int v = 0; v < 2 * n_; v++) {
            Arc *av = arc_a(v); const int nv = arc_n(v);
            if (nv == 0) continue;
            if (sdel_[v >> 1]) { for (int i = 0; i < nv; i++) { av[i].del = 1; n_red++; } continue; }
            for (int i = 0; i < nv; i++) mark[av[i].v] = 1;
            const int lmax = av[nv - 1].len + GAP_FUZZ;
            for (int i = 0; i < nv; i++) {
                if (mark[av[i].v] != 1) continue;
                const Arc *aw = arc_a(av[i].v); const int nw = arc_n(av[i].v);
                for (int j = 0; j < nw && aw[j].len + av[i].len <= lmax; j++) if (mark[aw[j].v]) mark[aw[j].v] = 2;
            }
            for (int i = 0; i < nv; i++) { if (mark[av[i].v] == 2) { av[i].del = 1; n_red++; } mark[av[i].v] = 0; }
        }
        if (n_red) { cleanup(); symm(); }
    }
    enum { ET_MERGEABLE = 0, ET_TIP = 1, ET_MULTI_OUT = 2, ET_MULTI_NEI = 3 };
    int is_utg_end(uint32_t v, uint32_t *lw) const
    {
        const Arc *av = arc_a(v ^ 1); const int nv0 = arc_n(v ^ 1);
        int i0 = -1, nv = 0, nw = 0;
        for (int i = 0; i < nv0; i++) if (!av[i].del) { i0 = i; nv++; }
        if (nv == 0) return ET_TIP;
        if (nv > 1) return ET_MULTI_OUT;
        if (lw) *lw = av[i0].v;
        const uint32_t w = av[i0].v ^ 1;
        const Arc *aw = arc_a(w); const int nw0 = arc_n(w);
        for (int i = 0; i < nw0; i++) if (!aw[i].del) nw++;
        return nw != 1 ? ET_MULTI_NEI : ET_MERGEABLE;
    }
    void cut_tip()
    {
        int cnt = 0;
        for (int v = 0; v < 2 * n_; v++) {
            if (sdel_[v >> 1]) continue;
            if (is_utg_end((uint32_t)v, nullptr) != ET_TIP) continue;
            uint32_t a[MAX_SHORT_TIP + 2], x = (uint32_t)v, lw = 0;
            int na = 0, max_ext = MAX_SHORT_TIP, ret;
            a[na++] = x;
            do {
                ret = is_utg_end(x ^ 1, &lw);
                if (ret != 0) break;
                a[na++] = lw;
                x = lw;
            } while (--max_ext > 0);
            if (ret == ET_MERGEABLE) continue;      // not a short dead end
            for (int i = 0; i < na; i++) seq_del(a[i] >> 1);
            cnt++;
        }
        if (cnt) cleanup();
    }

    // ---- unitig polishing
    bool edge_from_source(uint32_t query, uint32_t target, Arc &t) const
    {
        const int qn = (int)(query >> 1);
        for (int i = src_first_[qn]; i < src_first_[qn + 1]; i++) {      // (deleted hits too, as the reference)
            if (hit2arc(h_[i], t) < 0) continue;
            if (t.u == query && t.v == target) return true;
        }
        return false;
    }
    bool specific_edge(bool use_graph, uint32_t query, uint32_t target, Arc &t) const
    {
        if (use_graph) {
            const Arc *av = arc_a(query); const int nv = arc_n(query);
            for (int k = 0; k < nv; k++) if (!av[k].del && av[k].v == target) { t = av[k]; return true; }
        }
        return edge_from_source(query, target, t);
    }
    void overlap_len(int r, uint32_t &exact_len, uint32_t &inexact_len) const
    {
        exact_len = inexact_len = 0;
        for (int i = src_first_[r]; i < src_first_[r + 1]; i++) (h_[i].el == 1 ? exact_len : inexact_len) += (uint32_t)(h_[i].qe - h_[i].qs);
    }
    void reduce(std::vector<uint64_t> &u) const
    {
        u.erase(std::remove(u.begin(), u.end(), SKIP), u.end());
        for (size_t i = 0; i + 1 < u.size(); i++) {
            const uint32_t v = (uint32_t)(u[i] >> 32), w = (uint32_t)(u[i + 1] >> 32);
            Arc t; t.len = 0;
            const uint32_t l = specific_edge(true, v, w, t) ? (uint32_t)t.len : 0u;
            u[i] = (uint64_t)v << 32 | l;
        }
        if (!u.empty()) { const uint32_t v = (uint32_t)(u.back() >> 32); u.back() = (uint64_t)v << 32 | (uint32_t)len_[v >> 1]; }
    }
    void polish(std::vector<uint64_t> &u) const
    {
        const int n = (int)u.size();
        if (n < 3) return;
        uint32_t pre = (uint32_t)(u[0] >> 32);
        int pre_i = 0, skip = 0;
        for (int i = 1; i < n - 1; i++) {
            if (u[i] == SKIP) continue;
            const uint32_t v = (uint32_t)(u[i] >> 32);
            uint32_t afte = (uint32_t)(u[i + 1] >> 32);
            Arc pE, aE, t; memset(&pE, 0, sizeof(pE)); memset(&aE, 0, sizeof(aE));
            specific_edge(pre_i == i - 1, v ^ 1, pre ^ 1, pE);
            specific_edge(true, v, afte, aE);
            if (pE.el == 1 && aE.el == 1) { pre = v; pre_i = i; continue; }
            int afte_i = -1;
            for (int idx = i + 1; idx < n; idx++) {       // detect_exact_ovec: the next read that overlaps `pre` exactly
                if (!edge_from_source(pre, (uint32_t)(u[idx] >> 32), t)) break;
                if (t.el != 1) continue;
                afte_i = idx;
                break;
            }
            if (afte_i < 0) { pre = v; pre_i = i; continue; }
            afte = (uint32_t)(u[afte_i] >> 32);
            uint32_t el, il, min_inexact = 0xffffffffu, max_exact = 0;
            for (int k = i; k < afte_i; k++) { overlap_len((int)(u[k] >> 33), el, il); if (il < min_inexact) { min_inexact = il; max_exact = el; } }
            overlap_len((int)(pre >> 1), el, il);
            if (il > min_inexact || (il == min_inexact && el <= max_exact)) { pre = v; pre_i = i; continue; }
            overlap_len((int)(afte >> 1), el, il);
            if (il > min_inexact || (il == min_inexact && el <= max_exact)) { pre = v; pre_i = i; continue; }
            for (int k = i; k < afte_i; k++) { u[k] = SKIP; skip++; }
        }
        if (skip) reduce(u);
    }
    // how many of the unitig's earlier reads that cover the stretch of read cur in front of read next agree with it base for base
    int consensus_rate(const std::vector<uint64_t> &u, int cur_i, int next_i, int &r_match, int &r_total) const
    {
        r_match = r_total = 0;
        if (cur_i < 1) return -1;
        if (u[cur_i] == SKIP || u[next_i] == SKIP) return 0;
        const uint32_t v = (uint32_t)(u[cur_i] >> 32), w = (uint32_t)(u[next_i] >> 32);
        Arc t, e;
        if (!edge_from_source(v, w, t)) return -1;
        const int v_end = t.len - 1;
        for (int j = cur_i - 1; j >= 0; j--) {
            if (u[j] == SKIP) continue;
            const uint32_t wj = (uint32_t)(u[j] >> 32);
            if (!edge_from_source(wj, v, e)) break;
            const int w_beg = e.len, w_end = std::min(w_beg + v_end, len_[wj >> 1] - 1);
            if (w_end - w_beg != v_end) break;       // does not cover the whole stretch
            r_total++;
            if (e.el == 1) r_match++;
            else if (rb_.words) {
                bool same = true;
                for (int p = 0; p <= v_end && same; p++) same = rb_.base(v, p) == rb_.base(wj, w_beg + p);
                if (same) r_match++;
            }
        }
        return 1;
    }
    void polish_advance(std::vector<uint64_t> &u) const
    {
        const int n = (int)u.size();
        if (n < 3) return;
        int skip = 0, match_v, total_v;
        for (int i = 1; i < n - 1; i++) {
            if (consensus_rate(u, i, i + 1, match_v, total_v) != 1) continue;
            double match_rate = total_v == 0 ? 0 : (double)match_v / (double)total_v;
            if (match_v >= total_v * 0.5 && total_v > 0 && match_v > 0) continue;     // most reads support this one
            int max_i = i, match_max = match_v; double match_rate_max = match_rate;
            for (int k = i - 1; k >= 0; k--) {
                if (u[k] == SKIP) continue;
                if (consensus_rate(u, k, i + 1, match_v, total_v) < 0) break;
                if (total_v == 0) break;
                match_rate = (double)match_v / (double)total_v;
                if (match_rate > match_rate_max || (match_rate == match_rate_max && match_v > match_max)) { max_i = k; match_max = match_v; match_rate_max = match_rate; }
            }
            for (int k = max_i + 1; k <= i; k++) { if (u[k] == SKIP) continue; u[k] = SKIP; skip++; }
        }
        if (skip) reduce(u);
    }
};

} // namespace fsv_layout
