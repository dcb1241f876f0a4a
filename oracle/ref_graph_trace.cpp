// oracle/ref_graph_trace.cpp -- TEST INFRASTRUCTURE ONLY (diagnosis; never linked into the product, not used by any test).
//
// The *reference's own* graph code with a window into it: oracle/ref.mk compiles this file against hifiasm-0.14's sources where
// they lie under /root/reference (Overlaps.cpp is pulled in as a translation unit) and links the reference's other objects, giving
// oracle/_ref/hifiasm_trace -- the reference binary with build_string_graph_without_clean replaced by a driver that replays
// clean_graph's call sequence (Overlaps.cpp:27087-27280, :27352-27404) and prints, after every step, the arcs and reads that step
// removed, then the unitig ma_ug_gen makes before ma_ug_seq polishes it.
//     oracle/_ref/hifiasm_trace -f0 -o t.asm -t 1 PS1_hp1.fa 2> trace.log
// This is how round 3 found that on FocalSV-sized read sets none of the cleaning rounds touches the graph, and that what differed
// from the round-2 layout was outside them: overlaps below 500 bases, detect_chimeric_reads, and the polishing inside ma_ug_seq
// (oracle/layout.c, focalsv_amd/csrc/layout.h).
#define build_string_graph_without_clean build_string_graph_without_clean_ref
#include REF_OVERLAPS_CPP
#undef build_string_graph_without_clean
#include <set>
#include <string>
static std::set<std::string> g_prev;
static std::set<int> g_prev_del;
static void dump(asg_t *sg, const char *what)
{
    std::set<std::string> cur; std::set<int> del;
    if (sg) {
        for (uint32_t i = 0; i < sg->n_arc; i++) { asg_arc_t *a = &sg->arc[i]; if (a->del) continue; char b[128]; uint32_t v = a->ul >> 32;
            sprintf(b, "%u%c->%u%c ol=%u el=%u len=%u", v >> 1, "+-"[v & 1], a->v >> 1, "+-"[a->v & 1], a->ol, a->el, (uint32_t)a->ul); cur.insert(b); }
        for (uint32_t i = 0; i < sg->n_seq; i++) if (sg->seq[i].del) del.insert(i);
    }
    fprintf(stderr, "STEP %s: arcs %zu, deleted seqs %zu\n", what, cur.size(), del.size());
    for (auto &s : g_prev) if (!cur.count(s)) fprintf(stderr, "   - %s\n", s.c_str());
    for (auto &s : cur) if (!g_prev.count(s) && !g_prev.empty()) fprintf(stderr, "   + %s\n", s.c_str());
    for (int d : del) if (!g_prev_del.count(d)) fprintf(stderr, "   seq %d deleted\n", d);
    if (g_prev.empty()) for (auto &s : cur) fprintf(stderr, "   = %s\n", s.c_str());
    g_prev = cur; g_prev_del = del;
}
static void dump_hits(ma_hit_t_alloc *sources, long long n_read, ma_sub_t *cc, const char *what)
{
    long long n = 0, nd = 0;
    for (long long i = 0; i < n_read; i++) for (uint32_t j = 0; j < sources[i].length; j++) { if (sources[i].buffer[j].del) nd++; else n++; }
    fprintf(stderr, "HITS %s: live %lld del %lld\n", what, n, nd);
    if (cc) for (long long i = 0; i < n_read; i++) fprintf(stderr, "   read %lld cut [%u,%u) del %u\n", i, cc[i].s, cc[i].e, cc[i].del);
}
void build_string_graph_without_clean(int min_dp, ma_hit_t_alloc* sources, ma_hit_t_alloc* reverse_sources, long long n_read, uint64_t* readLen,
    long long mini_overlap_length, long long max_hang_length, long long clean_round, long long gap_fuzz, float min_ovlp_drop_ratio,
    float max_ovlp_drop_ratio, char* output_file_name, long long bubble_dist, int read_graph, int write)
{
    R_to_U ruIndex; init_R_to_U(&ruIndex, n_read);
    asg_t *sg = NULL; ma_sub_t* coverage_cut = NULL;
    min_thres = asm_opt.max_short_tip + 1;
    dump_hits(sources, n_read, NULL, "start");
    try_rescue_overlaps(sources, reverse_sources, n_read, 4);
    dump_hits(sources, n_read, NULL, "try_rescue_overlaps");
    renew_graph_init(sources, reverse_sources, sg, coverage_cut, &ruIndex, n_read);
    normalize_ma_hit_t_single_side_advance(sources, n_read);
    normalize_ma_hit_t_single_side_advance(reverse_sources, n_read);
    dump_hits(sources, n_read, NULL, "normalize");
    memset(R_INF.trio_flag, AMBIGU, R_INF.total_reads*sizeof(uint8_t));
    clean_weak_ma_hit_t(sources, reverse_sources, n_read);
    dump_hits(sources, n_read, NULL, "clean_weak");
    ma_hit_sub(min_dp, sources, n_read, readLen, mini_overlap_length, &coverage_cut);
    dump_hits(sources, n_read, coverage_cut, "ma_hit_sub");
    detect_chimeric_reads(sources, n_read, readLen, coverage_cut, asm_opt.max_ov_diff_final * 2.0);
    dump_hits(sources, n_read, NULL, "chimeric");
    ma_hit_cut(sources, n_read, readLen, mini_overlap_length, &coverage_cut);
    dump_hits(sources, n_read, NULL, "ma_hit_cut");
    ma_hit_flt(sources, n_read, coverage_cut, max_hang_length, mini_overlap_length);
    dump_hits(sources, n_read, NULL, "ma_hit_flt");
    ma_hit_contained_advance(sources, n_read, coverage_cut, &ruIndex, max_hang_length, mini_overlap_length);
    dump_hits(sources, n_read, coverage_cut, "contained");
    sg = ma_sg_gen(sources, n_read, coverage_cut, max_hang_length, mini_overlap_length);
    dump(sg, "ma_sg_gen");
    asg_arc_del_trans(sg, gap_fuzz); dump(sg, "del_trans");
    asm_opt.coverage = get_coverage(sources, coverage_cut, n_read);
    asg_cut_tip(sg, asm_opt.max_short_tip); dump(sg, "cut_tip");
    double cut_step = clean_round == 1 ? max_ovlp_drop_ratio : (max_ovlp_drop_ratio - min_ovlp_drop_ratio) / (clean_round - 1);
    double drop_ratio = min_ovlp_drop_ratio;
    for (int i = 0; i < clean_round; i++, drop_ratio += cut_step) {
        if (drop_ratio > max_ovlp_drop_ratio) drop_ratio = max_ovlp_drop_ratio;
        char nm[64];
        pre_clean(sources, coverage_cut, sg, bubble_dist); sprintf(nm, "r%d pre_clean", i); dump(sg, nm);
        asg_arc_identify_simple_bubbles_multi(sg, 1);
        asg_arc_del_false_node(sg, sources, asm_opt.max_short_tip); sprintf(nm, "r%d del_false_node", i); dump(sg, nm);
        asg_cut_tip(sg, asm_opt.max_short_tip); sprintf(nm, "r%d cut_tip a", i); dump(sg, nm);
        asg_arc_identify_simple_bubbles_multi(sg, 0);
        asg_arc_del_short_diploid_by_exact(sg, asm_opt.max_short_tip, sources); sprintf(nm, "r%d short_diploid_by_exact", i); dump(sg, nm);
        asg_cut_tip(sg, asm_opt.max_short_tip); sprintf(nm, "r%d cut_tip b", i); dump(sg, nm);
        asg_arc_identify_simple_bubbles_multi(sg, 1);
        asg_arc_del_short_diploid_by_length(sg, drop_ratio, asm_opt.max_short_tip, reverse_sources, asm_opt.max_short_tip, 1, 1, 0, 0, &ruIndex); sprintf(nm, "r%d by_length", i); dump(sg, nm);
        asg_cut_tip(sg, asm_opt.max_short_tip); sprintf(nm, "r%d cut_tip c", i); dump(sg, nm);
        asg_arc_identify_simple_bubbles_multi(sg, 1);
        asg_arc_del_short_false_link(sg, 0.6, 0.85, bubble_dist, reverse_sources, asm_opt.max_short_tip, &ruIndex); sprintf(nm, "r%d false_link", i); dump(sg, nm);
        asg_arc_identify_simple_bubbles_multi(sg, 1);
        asg_arc_del_complex_false_link(sg, 0.6, 0.85, bubble_dist, reverse_sources, asm_opt.max_short_tip); sprintf(nm, "r%d complex_false_link", i); dump(sg, nm);
        asg_cut_tip(sg, asm_opt.max_short_tip); sprintf(nm, "r%d cut_tip d", i); dump(sg, nm);
    }
    pre_clean(sources, coverage_cut, sg, bubble_dist); dump(sg, "final pre_clean");
    asg_arc_del_short_diploi_by_suspect_edge(sg, asm_opt.max_short_tip); dump(sg, "suspect_edge");
    asg_cut_tip(sg, asm_opt.max_short_tip); dump(sg, "cut_tip");
    asg_arc_del_triangular_directly(sg, asm_opt.max_short_tip, reverse_sources, &ruIndex); dump(sg, "triangular_directly");
    asg_arc_identify_simple_bubbles_multi(sg, 0);
    asg_arc_del_orthology_multiple_way(sg, reverse_sources, 0.4, asm_opt.max_short_tip, &ruIndex); dump(sg, "orthology");
    asg_cut_tip(sg, asm_opt.max_short_tip); dump(sg, "cut_tip");
    asg_arc_identify_simple_bubbles_multi(sg, 0);
    asg_arc_del_too_short_overlaps(sg, 2000, min_ovlp_drop_ratio, reverse_sources, asm_opt.max_short_tip, &ruIndex); dump(sg, "too_short_overlaps");
    asg_cut_tip(sg, asm_opt.max_short_tip); dump(sg, "cut_tip");
    asg_arc_del_simple_circle_untig(sources, coverage_cut, sg, 100, 0); dump(sg, "simple_circle");
    rescue_contained_reads_aggressive(NULL, sg, sources, coverage_cut, &ruIndex, max_hang_length, mini_overlap_length, bubble_dist, 10, 1, 0, NULL, NULL); dump(sg, "rescue_contained");
    rescue_missing_overlaps_aggressive(NULL, sg, sources, coverage_cut, &ruIndex, max_hang_length, mini_overlap_length, bubble_dist, 1, 0, NULL); dump(sg, "rescue_missing");
    rescue_missing_overlaps_backward(NULL, sg, sources, coverage_cut, &ruIndex, max_hang_length, mini_overlap_length, bubble_dist, 10, 1, 0); dump(sg, "rescue_backward");
    { ma_ug_t *ug = ma_ug_gen(sg); for (size_t u = 0; u < ug->u.n; u++) { fprintf(stderr, "UTG %zu len %u:", u, ug->u.a[u].len); for (uint32_t k = 0; k < ug->u.a[u].n; k++) fprintf(stderr, " %u%c(%u)", (uint32_t)(ug->u.a[u].a[k] >> 33), "+-"[(ug->u.a[u].a[k] >> 32) & 1], (uint32_t)ug->u.a[u].a[k]); fprintf(stderr, "\n"); } ma_ug_destroy(ug); }
    output_unitig_graph(sg, coverage_cut, output_file_name, sources, &ruIndex, max_hang_length, mini_overlap_length);
    output_contig_graph_primary_pre(sg, coverage_cut, output_file_name, sources, reverse_sources, asm_opt.small_pop_bubble_size, asm_opt.max_short_tip, &ruIndex, max_hang_length, mini_overlap_length);
    rescue_bubble_by_chain(sg, coverage_cut, sources, reverse_sources, bubble_dist, (asm_opt.max_short_tip*2), 0.15, 3, &ruIndex, 0.05, 0.9, max_hang_length, mini_overlap_length, 10, gap_fuzz);
    dump(sg, "rescue_bubble_by_chain");
    output_contig_graph_primary(sg, coverage_cut, output_file_name, sources, reverse_sources, bubble_dist, (asm_opt.max_short_tip*2), 0.15, 3, &ruIndex, 0.05, 0.9, max_hang_length, mini_overlap_length);
    dump(sg, "output_contig_graph_primary");
    asg_destroy(sg); free(coverage_cut); destory_R_to_U(&ruIndex);
}
