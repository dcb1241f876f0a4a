/* oracle/oracle.h -- TEST INFRASTRUCTURE ONLY: shared types of the CPU restatement. */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>

typedef struct {
    uint64_t hash;
    uint32_t pos;   /* index of the k-mer's last base in the read */
    uint8_t  rev;   /* strand with the smaller encoding */
    uint8_t  span;  /* uncompressed bases covered */
    uint16_t pad;
} orc_mz;           /* 16 bytes, same layout as fsv_mz */

#define ORC_WINDOW 375
#define ORC_K_FULL 15
#define ORC_K_MAX  31
#define ORC_PATH_CAP 512
#define ORC_PATH_REC 416   /* ops a path record of the HIP path holds (FSV_PATH_CAP) */
#define ORC_K_WIDE 127   /* widest band of the 256-bit restatement (oracle/bpm.c); the HIP path holds k <= 95 */

typedef struct {
    int32_t k, w, hpc;        /* minimizer scheme: 51, 51, 1 (hifiasm defaults, CommandLines.cpp:109-166) */
    int32_t n_rounds;         /* correction rounds: 3 */
    int32_t min_ovlp;         /* shortest overlap kept: 500 */
    int32_t min_anchors;      /* shortest chain kept: 3 */
    int32_t lookback;         /* chain DP predecessors examined: 64 */
    int32_t bw_ec;            /* chain indel budget per mille in correction rounds: 20 (0.02) */
    int32_t bw_final;         /* ... in the final overlap pass: 0 = co-linear anchors only (hifiasm: 0.001; a read that ends inside a
                                 homopolymer run yields an HPC k-mer whose end is off by the truncated bases, which a non-zero
                                 budget lets into the chain and shifts the exact-overlap interval by one) */
    int32_t min_contig_reads; /* chains of fewer reads are dropped (asg_cut_tip with max_short_tip = 3, Overlaps.cpp:4666): 4 */
    int32_t partition;        /* 1: haplotype partition of every read's overlaps before the consensus, as hifiasm does for every read set (default); 0: off (ONT profile) */
    /* error model (hifiasm: fixed for HiFi reads; raised for the ONT profile of BASELINE configs[4], where the reference has Flye) */
    int32_t win_rate_pm;      /* window threshold = x_len x this / 1000: 40 (max_ov_diff_ec 0.04 -> k = 15 for a full window) */
    int32_t k_cap;            /* largest threshold the rescue pass doubles to: 31 (THRESHOLD_MAX_SIZE, Hash_Table.h:9-22) */
    int32_t accept_err_pm;    /* an overlap is used when its error rate is at most this / 1000: 30 (Correct.cpp:725) */
    int32_t bw_rechain;       /* indel budget per mille when the final pass re-chains a pair without an exact overlap: 1 (max_ov_diff_final 0.001) */
    int32_t w_later;          /* minimizer window from the second correction round on (0 = w throughout, as hifiasm) */
    int32_t second_round;     /* 1: the junctions between grid windows get a second consensus (process_boundary, Correct.cpp:4453) */
    int32_t ins_dag;          /* 1: inserted strings that disagree go through hifiasm's DAG (build_DAGCon); 0: the most frequent string (ONT profile) */
    int32_t min_anchors_final;/* shortest chain of the final overlap pass: 1 -- hifiasm keeps every (target, strand) group that shares a minimizer
                                 (calculate_overlap_region_by_chaining, Hash_Table.cpp:684-745: no minimum) */
    int32_t min_ovlp_final;   /* shortest final overlap: 1 (the graph drops what is below 50 bases: ma_hit_cut) */
    int32_t graph_layout;     /* 1: the layout as hifiasm's string graph + unitig polishing (oracle/layout.c); 0: best-buddy chains (ONT profile) */
    int32_t left_rescue;      /* 1: the rescue pass also walks left from a matched window (recalcate_window_advance, Correct.cpp:2745-2905) */
    int32_t junction_cigars;  /* 1: the haplotype partition reads the columns beside a window junction off the re-aligned junction cigar
                                 (calculate_boundary_cigars, Correct.cpp:2310; markSNP_advance :5054) */
    int32_t fix_boundary;     /* 1: a window whose alignment touches the edge of its band is aligned once more with the band shifted (fix_boundary, Correct.cpp:1676) */
    int32_t partial_charge;   /* 1: an unmatched window beside a matched one is charged what two extension alignments leave uncovered (non_trim_error_rate,
                               * Correct.cpp:725-845) instead of its length.  ORACLE ONLY so far (the HIP path charges the length): default 0 */
} orc_asm_params;

typedef struct {
    uint32_t q, t;            /* read indices inside the set */
    int32_t x_s, x_e;         /* inclusive range on q (forward strand) */
    int32_t y_s, y_e;         /* inclusive range on t in strand coordinates */
    int32_t score, n_chain;
    int32_t chain_off;        /* offset of this overlap's anchors in the chain arrays */
    int32_t first_win, n_win;
    int32_t align_len, err_sum;
    uint8_t rev, is_match, exact, pad;
} orc_ovl;                    /* 56 bytes, same layout as fsv_ovl */

typedef struct {
    uint32_t ovl, win;
    int32_t x_start, y_start; /* y_start: predicted partner of x_start (before the -k pad) */
    int32_t y_beg;            /* first in-read base covered by the padded window */
    int32_t end_site, err;
    int32_t ry_start, ry_end; /* absolute strand coordinates of the aligned y interval (after S6) */
    int16_t x_len, extra_begin, extra_end, path_len;
    uint8_t k, rescued, pad[2];
    uint8_t path[ORC_PATH_CAP]; /* start-to-end ops: 0 match 1 mismatch 2 y-only 3 x-only */
} orc_win;

int orc_thr_for_len(int x_len);
int orc_double_thr(int pre, int x_len);
int orc_thr_for_len_p(const orc_asm_params *P, int x_len);
int orc_double_thr_p(const orc_asm_params *P, int pre, int x_len);
int orc_unique_sorted(orc_mz *mz, int n);
int orc_chain_pair(const orc_mz *q, int nq, int lenq, const orc_mz *t, int nt, int lent, const orc_asm_params *P,
                   int bw_per_mille, orc_ovl *o, int32_t *chain_qe, int32_t *chain_te, int chain_cap);
void orc_asm_default_params(orc_asm_params *P);
int orc_layout_graph(const char *const *seq, const int *len, int n, const orc_ovl *hit, int n_hit, int min_reads, int32_t *piece_read,
                     uint8_t *piece_rev, int32_t *piece_len, int32_t *contig_first, int piece_cap, int contig_cap);
/* whole path for one read set; see oracle/asm.c */
int orc_assemble(const char *seqs, const uint64_t *seq_off, int n_reads, const orc_asm_params *P,
                 char *contigs, uint64_t contigs_cap, uint64_t *contig_off, int contig_cap, int *n_contigs,
                 char *corrected, uint64_t corrected_cap, uint64_t *corrected_off);
typedef struct {
    int32_t k, w;             /* seeds: 19, 19 (minimap2 asm5); callers raise w to len/3000+1 for long windows */
    int32_t min_anchors, lookback, max_gap;  /* max_gap 50 000: a chain runs across any SV DipPAV would call (max_svlen, extract_contig_signature_CCS.py:411) */
    int32_t a, b, q, e, q2, e2; /* asm5: 1, 19, 39, 3, 81, 1; q2 < 0 = single affine */
    int32_t pad;              /* identical bases added on each side of a DP event: 24 */
    int32_t max_mm_run;       /* an equal-length inter-seed run with at most this many mismatches stays 'M': 4 */
    int32_t xdrop;            /* gap-free end extension: 100 */
    int32_t max_cells;        /* largest DP event: 2^26 cells; a larger one is seeded again on its own (oracle/aln.c: sub_align) */
} orc_aln_params;

typedef struct {
    int32_t ref_start, ref_end; /* 0-based, end exclusive (pysam pos / reference_end) */
    int32_t q_start, q_end;     /* aligned part of the (strand-oriented) contig */
    int32_t n_cigar, n_chain;
    uint8_t rev, mapq, pad[2];
} orc_aln;

void orc_aln_default_params(orc_aln_params *P);
int orc_nw(const char *t, int tl, const char *q, int ql, const orc_aln_params *P, uint32_t *cigar, int cigar_cap, int *n_cigar, uint8_t *bt);
int orc_aln_chain(const orc_mz *mq, int nq, int lenq, const orc_mz *mt, int nt, const orc_aln_params *P, int *rev_out,
                  int32_t *cq, int32_t *ct, int cap);
/* longest left shift (< = cap) of a gap of `len` bases that starts at s[off]: the count of l with s[off-1-l] == s[off+len-1-l] */
int orc_gap_max_shift(const char *s, int off, int len, int cap);
int orc_align_contig(const char *contig, int lenq, const char *ref, int lent, const orc_aln_params *P, orc_aln *out,
                     uint32_t *cigar, int cigar_cap);
#define ORC_ALN_MAX_REC 5      /* records per contig: primary + supplementary chains of its strand, the part behind a cut, one chain of the other strand */
#define ORC_ALN_MAJ_REC 3      /* chains on the contig's majority strand */
#define ORC_ALN_SUP_MIN 200    /* chain score a supplementary chain needs */
#define ORC_ALN_SUB_OCC 2      /* an event larger than max_cells is seeded again: minimizers occurring at most this often in each side */
#define ORC_ALN_SUB_PER 1500   /* ... with a minimizer window of max(w, L / 1500 + 1) for a box whose longer side has L bases */
#define ORC_ALN_AMAX 8192      /* anchors the chaining tile of the HIP path holds */
int orc_occ_sorted(orc_mz *mz, int n, int max_occ);
int orc_aln_chains(const orc_mz *mq, int nq, int lenq, const orc_mz *mt, int nt, const orc_aln_params *P, int *rev_out,
                   int32_t *cq, int32_t *ct, int cap, int *chain_n, uint8_t *chain_rev, int max_rec);
int orc_align_contig_multi(const char *contig, int lenq, const char *ref, int lent, const orc_aln_params *P, orc_aln *out,
                           uint32_t *cigar, int cigar_cap, int max_rec);
int orc_sketch(const char *s, int len, int w, int k, int hpc, orc_mz *out, int cap);
int orc_bpm(const char *y, int m, const char *x, int n, int k, int *err);
int orc_bpm_extension(const char *y, const char *x, int n, int k, int *err, int *p_end);
int orc_bpm_wide(const char *y, int m, const char *x, int n, int k, int *err);
int orc_bpm_path_wide(const char *y, int m, const char *x, int n, int k, int *err, int *start_site, int *path_len, uint8_t *path, uint64_t *cols);
int orc_banded_dp_plain(const char *y, int m, const char *x, int n, int k, uint8_t *ends);
int orc_bpm_path(const char *y, int m, const char *x, int n, int k, int *err, int *start_site, int *path_len,
                 uint8_t *path, uint64_t *cols);
int orc_try_cigar(const char *y, const char *x, int n, int end_site, int error, uint8_t *path, int *start_site, int *path_len);
int orc_generate_cigar(uint8_t *path, int plen, int n, const char *x, const char *y, int *start, int *end, int *err,
                       int *run_len, uint8_t *run_op);
#endif
