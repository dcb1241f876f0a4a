/* focalsv_hip.h -- C ABI of libfocalsv_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for FocalSV's per-region local-assembly +
 * SV-calling hot path.  The reference has no plugin API: its boundary is
 * "spawn a CPU binary and read its files back" (SURVEY.md 8b):
 *
 *   hifiasm -o <prefix> -t T <reads.fa>        focalsv/3_assembly/run_assembly.py:15-26
 *       -> GFA 'S' lines -> HP1.fa/HP2.fa       focalsv/3_assembly/post_assembly.py:79-95, combine_fas.py:10-35
 *   minimap2 -a -x asm5 --cs -r2k ref asm.fa   focalsv/4_sv_calling/Dippav/DipPAV_variant_call.py:103-108
 *       -> BAM records read through pysam       focalsv/4_sv_calling/Dippav/extract_contig_signature_CCS.py:14-47,342-432
 *
 * The entry points below are what a binding at those two call sites would
 * bind instead (INTEGRATION.md shows the ctypes stubs).  Conventions:
 *   - plain C types, pointers + sizes, no ownership transfer: every buffer is
 *     allocated by the caller (sized through the *_bound() calls) ;
 *   - every function returns 0 or a negative FSV_E* code, never aborts; batch
 *     calls also fill a per-unit status array;
 *   - one fsv_ctx per GPU/stream; calls on different contexts are independent,
 *     the library keeps no global state;
 *   - "_dev" arguments are device pointers valid on the context's device
 *     (hipMalloc / torch tensor data_ptr()); everything else is host memory.
 *   - there is NO CPU fallback: without a usable gfx950 device
 *     fsv_ctx_create() fails with FSV_ENODEV.
 *
 * Read store layout (hifiasm K0, Process_Read.h:108-137, restated for HBM):
 * bases are 2-bit codes A=0 C=1 G=2 T=3, 16 per little-endian uint32 word,
 * base i of a read in bits [2*(i%16), 2*(i%16)+1] of word (word_off + i/16).
 * Every read starts on a word boundary.  In a READ 'N' is stored as A (hifiasm keeps an
 * N side list; the FocalSV read FASTAs come from BAM records and carry none); the aligner's
 * reference windows keep their N (fsv_align_batch).
 */
#ifndef FOCALSV_HIP_H
#define FOCALSV_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSV_OK        0
#define FSV_ENODEV   -1  /* no gfx950 device / HIP runtime unusable */
#define FSV_EINVAL   -2  /* bad argument */
#define FSV_ENOMEM   -3  /* device or host allocation failed */
#define FSV_EHIP     -4  /* HIP runtime error (fsv_last_error gives the text) */
#define FSV_ECAP     -5  /* caller buffer too small */
#define FSV_EUNSUP   -6  /* input outside what this build supports */
#define FSV_EINTERNAL -7 /* an invariant of the library did not hold, or a C++ exception was caught at the boundary (fsv_last_error gives the text) */

#define FSV_WINDOW          375 /* hifiasm WINDOW, Hash_Table.h:9 */
#define FSV_K_FULL           15 /* hifiasm THRESHOLD, Hash_Table.h:13 */
#define FSV_K_MAX            31 /* hifiasm THRESHOLD_MAX_SIZE, Hash_Table.h:17 */
#define FSV_K_WIDE           95 /* widest threshold of the wide-band kernels (191 rows in six 32-bit limbs): ONT-profile reads */

typedef struct fsv_ctx fsv_ctx;

/* ---- context ----------------------------------------------------------- */
int  fsv_ctx_create(int device, fsv_ctx **out);
void fsv_ctx_destroy(fsv_ctx *ctx);
/* use an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream); 0 = the context's own */
int  fsv_ctx_set_stream(fsv_ctx *ctx, void *hip_stream);
int  fsv_ctx_sync(fsv_ctx *ctx);
const char *fsv_last_error(const fsv_ctx *ctx);
const char *fsv_strerror(int code);
int  fsv_version(void);
/* device properties the benchmarks print: CU count, clock (kHz), HBM bytes */
int  fsv_device_info(const fsv_ctx *ctx, int *n_cu, int *clock_khz, uint64_t *hbm_bytes, char *name, size_t name_cap);

/* raw device memory helpers for callers without torch */
int  fsv_dev_alloc(fsv_ctx *ctx, size_t bytes, void **dev_ptr);
int  fsv_dev_free(fsv_ctx *ctx, void *dev_ptr);
int  fsv_h2d(fsv_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
int  fsv_d2h(fsv_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);

/* ---- K0: read store ------------------------------------------------------
 * Replaces hifiasm's FASTA ingest + ha_compress_base (Assembly.cpp:21-65).
 * Host-side packing of ASCII reads into the layout above.
 *   seqs      concatenated ASCII bases of all reads
 *   seq_off   n_reads+1 offsets into seqs
 *   words     out, capacity words_cap (>= fsv_pack_bound)
 *   word_off  out, n_reads+1 word offsets (read r occupies [word_off[r], word_off[r+1]))
 */
size_t fsv_pack_bound(const uint64_t *seq_off, uint32_t n_reads);
int    fsv_pack_reads(const char *seqs, const uint64_t *seq_off, uint32_t n_reads,
                      uint32_t *words, size_t words_cap, uint64_t *word_off);

/* ---- K5: batched banded bit-parallel edit distance -------------------------
 * Replaces Reserve_Banded_BPM / Reserve_Banded_BPM_4_SSE_only as driven by
 * verify_window (hifiasm-0.14 Levenshtein_distance.h:274-461, 893-1198;
 * Correct.cpp:203-250, 306-531).  One task = one (x window, overlapping read) pair.
 */
typedef struct fsv_wtask {
    uint32_t x_word;   /* word offset of read x in the store */
    uint32_t y_word;   /* word offset of read y */
    int32_t  x_start;  /* first base of the window in x (forward strand) */
    int32_t  y_start;  /* chain-predicted partner of x_start in y's strand coordinates (before the -k pad) */
    int32_t  y_len;    /* length of read y */
    uint16_t x_len;    /* 1..FSV_WINDOW */
    uint8_t  k;        /* error threshold, band = 2k+1, <= FSV_K_MAX (hifiasm) or <= FSV_K_WIDE (wide-band kernels) */
    uint8_t  y_rev;    /* 1: y is read on its reverse-complement strand */
    uint32_t ovl;      /* caller tag: overlap id */
    uint32_t win;      /* caller tag: window index inside the overlap */
} fsv_wtask;           /* 32 bytes */

typedef struct fsv_wres {
    int32_t end_site;    /* 0-based end offset inside the padded y window, -1 = no alignment within k */
    int32_t err;         /* edit distance, -1 = none */
    int32_t y_beg;       /* first y base covered by the padded window after clipping to the read (determine_overlap_region) */
    int16_t extra_begin; /* 'N' columns padded in front (window starts before the read) */
    int16_t extra_end;   /* 'N' columns padded behind */
} fsv_wres;              /* 16 bytes */

/* K6 result of one window task: the alignment path after generate_cigar's end trimming and gap left-shift
 * (Reserve_Banded_BPM_PATH + generate_cigar, Levenshtein_distance.h:511-888, Correct.cpp:1387-1536). */
typedef struct fsv_wpath {
    int32_t ry_start, ry_end;   /* aligned y interval in y's strand coordinates (inclusive) */
    int16_t path_len, err;      /* ops in the path; edit distance after trimming */
    uint8_t state;              /* 0 no alignment within k, 1 path present */
    uint8_t y_rev;
    uint16_t pad;
    uint32_t y_word;
    int32_t y_len;
    uint8_t ops[104];           /* 2 bits per op, start-to-end: 0 match 1 mismatch 2 y-only 3 x-only; at most 416 ops */
} fsv_wpath;                    /* 128 bytes */

/* K5 + K6 over host tasks (test / integration entry point; fsv_assemble_batch runs the same kernels on its own task list):
 * res[i] as fsv_bpm_windows, paths[i] for every task with res[i].err >= 0. */
int fsv_bpm_paths(fsv_ctx *ctx, const uint32_t *store, size_t store_words, const fsv_wtask *tasks,
                  uint32_t n_tasks, fsv_wres *res, fsv_wpath *paths);

int fsv_bpm_windows_dev(fsv_ctx *ctx, const uint32_t *store_dev, const fsv_wtask *tasks_dev,
                        uint32_t n_tasks, fsv_wres *res_dev);
/* host convenience wrapper (copies in, runs, copies out) */
int fsv_bpm_windows(fsv_ctx *ctx, const uint32_t *store, size_t store_words, const fsv_wtask *tasks,
                    uint32_t n_tasks, fsv_wres *res);

/* ---- assembler boundary -----------------------------------------------------
 * Replaces the process boundary `hifiasm -o <prefix> -t <T> <reads.fa>` and the GFA
 * read-back (focalsv/3_assembly/run_assembly.py:15-44, post_assembly.py:79-95).
 * One "read set" = one FASTA the reference would hand to one hifiasm process
 * (PS<ps>_hp1.fa, PS<ps>_hp2.fa of a region).  Many sets are assembled per call: a
 * single 50 kb region cannot fill 256 CUs.
 *
 * Stages (hifiasm-0.14 file:line in DESIGN.md): minimizer sketch -> per-read index ->
 * anchors + chaining -> 375-bp window verification (K5) -> rescue/accept -> window
 * paths (K6) -> per-column consensus (K8) -> re-pack + reverse-complement, n_rounds
 * times; then exact overlaps, layout, contig stitching.
 */
typedef struct fsv_asm_params {
    int32_t k, w, hpc;        /* minimizer scheme; hifiasm defaults 51, 51, 1 (CommandLines.cpp:109-166) */
    int32_t n_rounds;         /* correction rounds, 3 */
    int32_t min_ovlp;         /* shortest overlap kept in a correction round: 1 -- hifiasm keeps every (target, strand) group that shares a minimizer,
                               * however short (calculate_overlap_region_by_chaining, Hash_Table.cpp:684-745); the ONT / CLR profiles: 500 */
    int32_t min_anchors;      /* shortest chain kept: 1 (the same); the ONT / CLR profiles: 3 */
    int32_t lookback;         /* chain DP predecessors examined, 64 (= one wavefront) */
    int32_t bw_ec;            /* chain indel budget per mille in correction rounds, 20 (hifiasm 0.02) */
    int32_t bw_final;         /* ... in the final overlap pass, 0 = co-linear anchors only */
    int32_t min_contig_reads; /* chains of fewer reads are dropped, as hifiasm's asg_cut_tip(max_short_tip = 3) does (Overlaps.cpp:4666): 4 */
    /* error model: hifiasm's constants for HiFi reads; fsv_asm_ont_params() raises them for ONT-profile reads (10 % error), where the
     * reference runs Flye / Shasta instead (run_assembly.py:74-100) */
    int32_t win_rate_pm;      /* window threshold = x_len x this / 1000: 40 (max_ov_diff_ec 0.04: k = 15 = FSV_K_FULL for a full window) */
    int32_t k_cap;            /* largest threshold the rescue pass doubles to: 31 = FSV_K_MAX (THRESHOLD_MAX_SIZE); at most FSV_K_WIDE */
    int32_t accept_err_pm;    /* an overlap is used when its error rate is at most this / 1000: 30 (Correct.cpp:725) */
    int32_t bw_rechain;       /* indel budget per mille when the final pass re-chains a pair without an exact overlap: 1 (max_ov_diff_final 0.001) */
    int32_t w_later;          /* minimizer window from the second correction round on; 0 = w throughout (hifiasm) */
    int32_t partition;        /* 1: haplotype partition of every read's overlaps before the consensus, as hifiasm (partition_overlaps_advance,
                               * Correct.cpp:7127); 0: off (the ONT profile: at 10 % error coincident errors pass for alleles) */
    int32_t second_round;     /* 1 (default): hifiasm's second consensus pass over the window junctions (process_boundary, Correct.cpp:4453): K5 + K6 +
                               * consensus once more per junction.  0: a vote on the bases both window alignments skip at a junction stands in for
                               * it -- a quarter faster, same reads after three rounds on all but one of 7 800 golden reads (ONT profile: 0) */
    int32_t ins_dag;          /* 1 (default): inserted strings that disagree go through hifiasm's DAG of inserted strings (build_DAGCon, Correct.cpp:3893);
                               * 0: the most frequent string is inserted (ONT profile) */
    int32_t min_anchors_final;/* shortest chain of the final overlap pass: 1 -- hifiasm keeps every (target, strand) group that shares a minimizer
                               * (calculate_overlap_region_by_chaining, Hash_Table.cpp:684-745: no minimum); 0 = min_anchors */
    int32_t min_ovlp_final;   /* shortest final overlap: 1 (the graph drops what is below 50 bases, ma_hit_cut, Overlaps.cpp:1785); 0 = min_ovlp */
    int32_t graph_layout;     /* 1 (default): the layout as hifiasm-0.14 makes it -- chimeric-read detection, containment in read order, string graph with
                               * transitive reduction and tip cutting, unitig polishing (Overlaps.cpp:1698, 1031, 2152, 4531, 4666, 7759, 8480, 8893) --
                               * for every set that is not flagged FSV_SET_UNPHASED; 0: best-buddy chains (ONT profile, unphased sets) */
    int32_t junction_cigars;  /* 1 (default): the haplotype partition reads the ~50 columns on each side of a window junction off the re-aligned
                               * junction cigar, as hifiasm does (calculate_boundary_cigars, Correct.cpp:2310; markSNP_advance :5054); 0: window cigars */
} fsv_asm_params;
void fsv_asm_default_params(fsv_asm_params *p);
/* ONT-profile reads (BASELINE configs[4]: ~10 % error): k = 15, w = 15 without homopolymer compression (a 30 kb read then has ~3 750 minimizers: below the 4 096 a list holds), chain indel budget 0.15 / 0.05,
 * windows up to 25 % apart (k = 93: wide-band K5 / K6), overlaps up to 30 % error.  Parity unpinned: the reference has Flye here. */
void fsv_asm_ont_params(fsv_asm_params *p);
/* CLR reads (~12 % error, insertion-rich; the reference: flye --pacbio-raw, run_assembly.py:46-72): the ONT profile's values under a name of
 * their own.  Parity unpinned (no Flye in the tree or the image); planted truth in tests/test_gpu_ont.py.  Reads above ~32 kb lose the anchors
 * beyond their first 4 096 minimizers in either profile (set status bits 1 / 2 say so). */
void fsv_asm_clr_params(fsv_asm_params *p);

typedef struct fsv_mz {       /* ha_mz1_t, htab.h:8-13 */
    uint64_t hash;
    uint32_t pos;             /* index of the k-mer's last base */
    uint8_t  rev, span;
    uint16_t pad;
} fsv_mz;                     /* 16 bytes */

typedef struct fsv_ovl {      /* overlap_region / ma_hit_t, Hash_Table.h:67-101, Overlaps.h:56-64 */
    uint32_t q, t;            /* read indices inside the set */
    int32_t x_s, x_e;         /* inclusive range on q (forward strand) */
    int32_t y_s, y_e;         /* inclusive range on t, strand coordinates */
    int32_t score, n_chain;
    int32_t chain_off;        /* unused on the device */
    int32_t first_win, n_win; /* window tasks of this overlap */
    int32_t align_len, err_sum;
    uint8_t rev, is_match, exact, valid;
} fsv_ovl;                    /* 56 bytes */

typedef struct fsv_readsets {
    const uint32_t *store_dev;  /* device: 2-bit read store */
    const uint64_t *word_off;   /* host: n_reads+1 word offsets into the store */
    const int32_t  *read_len;   /* host: n_reads lengths in bases */
    const uint32_t *set_start;  /* host: n_sets+1 read indices; set s owns reads [set_start[s], set_start[s+1]) */
    uint32_t n_reads, n_sets;
    const uint8_t  *set_flags;  /* host: n_sets flags, or NULL; informational since round 2 (see FSV_SET_UNPHASED) */
} fsv_readsets;
#define FSV_SET_UNPHASED 1      /* unphased.fa (run_assembly.py:17-21).  The haplotype partition that keeps the other allele's overlaps out
                                 * of a read's consensus (partition_overlaps_advance, Correct.cpp:7127) runs for EVERY set, as in hifiasm,
                                 * so the flag no longer changes the result; callers may keep passing it */

typedef struct fsv_contigs {
    char     *seq;         /* host, capacity seq_cap: ASCII contig bases back to back */
    uint64_t  seq_cap;
    uint64_t *off;         /* host, capacity contig_cap+1 */
    uint32_t *set;         /* host, capacity contig_cap: owning set of each contig */
    uint32_t *n_reads;     /* host, capacity contig_cap: reads laid out in each contig */
    uint32_t  contig_cap;
    uint32_t  n_contigs;   /* out */
    int32_t  *set_status;  /* host, n_sets: 0 ok, >0 warning bits (FSV_W_*), <0 FSV_E* for that set only */
} fsv_contigs;

#define FSV_W_MZ_TRUNC     1  /* a read had more minimizers than the per-read cap; the rest were ignored */
#define FSV_W_ANCHOR_TRUNC 2  /* a read pair had more anchors than the chaining tile holds */
#define FSV_W_NO_LAYOUT    4  /* no chain of min_contig_reads reads: the set has no contig (hifiasm writes an empty GFA for it too) */
#define FSV_W_INS_EVENTS   8  /* a consensus window saw more inserted-base events than its buffer holds; the extra votes were dropped */
#define FSV_W_WINDOW_KEPT 16  /* a corrected window would have outgrown its slot; the read keeps that window uncorrected */
#define FSV_W_INTERNAL    32  /* a minimizer slot overflowed (cannot happen: one minimizer per base at most) */
#define FSV_W_SITES       64  /* haplotype partition: more than 255 candidate sites in a window, 1 024 in a read (512 once sites beside another site are
                                 * dropped), or the site pool ran out;
                                 * that window's sites / that read's partition were skipped */

/* capacity needed for fsv_contigs.seq / contig count for these read sets */
int fsv_assemble_batch_bound(const fsv_readsets *sets, uint64_t *seq_cap, uint32_t *contig_cap);
int fsv_assemble_batch(fsv_ctx *ctx, const fsv_readsets *sets, const fsv_asm_params *params, fsv_contigs *out);

/* Optional per-stage counters of the last fsv_assemble_batch on this context (for the roofline accounting):
 * window tasks verified (K5), paths computed (K6), DP column steps, algorithmic bytes (SURVEY.md 8d model). */
#define FSV_MAX_KERNEL_STATS 16
typedef struct fsv_kernel_stat {
    char     name[24];
    double   ms;           /* summed HIP-event time of this kernel's launches on the context's stream */
    uint64_t launches;
    uint64_t algo_bytes;   /* compulsory bytes in + out of those launches (per-kernel table) */
} fsv_kernel_stat;

typedef struct fsv_asm_stats {
    uint64_t n_pairs, n_overlaps, n_windows, n_windows_matched, n_paths, n_path_dp;
    uint64_t dp_columns;       /* K5 + K6 column steps (windows x their x_len) */
    uint64_t algo_bytes;       /* packed operand + result bytes of all DP tasks + reads in + contigs out */
    uint64_t n_exact_overlaps;  /* overlaps handed to the layout (exact, or inexact ones the last correction round verified) */
    uint64_t n_inexact_candidates; /* pairs re-chained with the gapped bandwidth in the final pass */
    uint64_t n_path_fr;         /* of those, distance <= 3: walked without the DP matrix (k_path_fr) */
    uint64_t n_junction_cigars; /* junctions re-aligned for the haplotype partition (k_bcig_tasks) */
    uint64_t n_junction_used;   /* of those, cigars the partition reads (accepted, and showing something the window cigars do not) */
    double   ms_sketch, ms_chain, ms_verify, ms_path, ms_consensus, ms_final, ms_total;
    uint32_t n_kernels, pad;
    fsv_kernel_stat kernels[FSV_MAX_KERNEL_STATS];
} fsv_asm_stats;
int fsv_asm_last_stats(const fsv_ctx *ctx, fsv_asm_stats *out);

/* Test hook: after fsv_assemble_batch(...) with n_rounds = r, the corrected reads of the last round
 * can be fetched as ASCII (same order as the input reads). */
int fsv_asm_fetch_reads(fsv_ctx *ctx, char *seq, uint64_t seq_cap, uint64_t *off, uint32_t n_reads);

/* ---- K1 exposed: minimizer sketch of every read (ha_sketch, sketch.cpp:39-137) -------------------------
 * out_mz receives, per read, its minimizers in position order; out_off (n_reads+1) indexes them.
 * variant: 0 = library's choice (position-parallel kernel for odd k, deque replay otherwise), 1 = force the replay kernel. */
int fsv_sketch_reads(fsv_ctx *ctx, const fsv_readsets *sets, int32_t w, int32_t k, int32_t hpc, int32_t variant,
                     fsv_mz *out_mz, uint64_t out_cap, uint64_t *out_off);

/* ---- aligner boundary ---------------------------------------------------------
 * Replaces `minimap2 -a -x asm5 --cs -r2k ref_chr.fa assemblies.fa | samtools sort` and the pysam read-back
 * (focalsv/4_sv_calling/Dippav/DipPAV_variant_call.py:103-112; fields consumed by
 * extract_contig_signature_CCS.py:14-47, 279-282, 347-356: reference_name, pos, reference_end, cigar with
 * ops {0 M, 1 I, 2 D, 4 S, 5 H}, qname, is_reverse, mapq).  Every contig is aligned against the reference
 * window of its own region (the caller adds the window's chromosome offset to ref_start/ref_end).
 * minimap2 itself is not part of the reference tree; scoring is its published asm5 preset and the DP
 * follows the in-tree ksw2 (software/hifiasm-0.14/ksw2_extz2_sse.c, ksw2.h:115-150).
 */
typedef struct fsv_aln_params {
    int32_t k, w;               /* seeds: 19, 19; w grows with the sequence length (len/3000 + 1) */
    int32_t min_anchors, lookback, max_gap;  /* 3, 64, 50000: one chain runs across any SV DipPAV calls (max_svlen, extract_contig_signature_CCS.py:411) */
    int32_t a, b, q, e, q2, e2; /* asm5: 1, 19, 39, 3, 81, 1 */
    int32_t pad;                /* identical bases added on each side of a DP event, 24 */
    int32_t max_mm_run;         /* equal-length inter-seed run with <= this many mismatches stays M, 4 */
    int32_t xdrop;              /* gap-free end extension, 100 */
    int32_t max_cells;          /* largest DP event, 2^26 cells; a larger one (both copies of a duplication between two unique seeds) is seeded
                                   again on its own, and what is still larger inside it is closed from its corners: never refused */
} fsv_aln_params;
void fsv_aln_default_params(fsv_aln_params *p);

typedef struct fsv_aln_rec {
    int32_t  ref_start, ref_end;  /* 0-based in the window, end exclusive (pysam pos / reference_end) */
    int32_t  q_start, q_end;      /* aligned part of the strand-oriented contig */
    uint32_t n_cigar;
    uint32_t n_chain;
    uint64_t cigar_off;           /* into fsv_alns.cigar */
    uint32_t contig;              /* index of the contig */
    uint8_t  rev, mapq, pad[2];
} fsv_aln_rec;                    /* 40 bytes */

typedef struct fsv_alns {
    fsv_aln_rec *rec;       /* host, capacity rec_cap: 5 x n_contigs holds every case (FSV_ALN_MAX_REC records per contig) */
    uint32_t  rec_cap, n_rec;
    uint32_t *cigar;        /* host, BAM encoding len << 4 | op */
    uint64_t  cigar_cap, n_cigar;
    int32_t  *contig_status; /* host, n_contigs: 0 aligned, 1 no chain (unaligned), <0 FSV_E* for that contig */
} fsv_alns;

#define FSV_ALN_MAX_REC 5
/* contig i is aligned to reference window contig_ref[i].  A contig yields its primary record and, when the primary chain leaves
 * part of the contig uncovered (an SV beyond max_gap), up to two supplementary records on the same strand -- consecutive in
 * `rec`, primary first, soft clips for the parts the record does not align -- from which DipPAV's split-alignment rules call the
 * SV (extract_contig_signature_CCS.py:251-327).  An inverted piece of the contig comes back as one record of the other strand
 * (rev differs), and the record it interrupts is cut in two around it: the split rule pairs records of one strand only (:286),
 * so an inversion yields no INS / DEL call, as with minimap2's reverse-strand supplementary alignment.
 * contig_seq == NULL (contig_off ignored): align the n_contigs contigs of the last fsv_assemble_batch on this context straight
 * from device memory, in their output order -- the device-resident hand-off between the two boundaries.
 * Reference windows may hold N (anything but ACGT / acgt): an N pairs with nothing, costs 1 in the alignment score (minimap2's sc_ambi)
 * and does not seed; the alignment runs through a short run of N as 'M'. */
int fsv_align_batch(fsv_ctx *ctx, const char *contig_seq, const uint64_t *contig_off, uint32_t n_contigs,
                    const uint32_t *contig_ref, const char *ref_seq, const uint64_t *ref_off, uint32_t n_refs,
                    const fsv_aln_params *params, fsv_alns *out);

typedef struct fsv_aln_stats {
    uint64_t n_pairs, n_events, dp_cells, algo_bytes;
    double ms_seed, ms_chain, ms_events, ms_dp, ms_total;
    uint64_t n_boxes;       /* events larger than max_cells that were seeded again */
} fsv_aln_stats;
int fsv_aln_last_stats(const fsv_ctx *ctx, fsv_aln_stats *out);

/* single global alignment (the DP of one event), exposed for known-answer tests against ksw2 */
int fsv_nw(fsv_ctx *ctx, const char *target, int32_t tl, const char *query, int32_t ql, const fsv_aln_params *params,
           int32_t *score, uint32_t *cigar, uint32_t cigar_cap, uint32_t *n_cigar);

/* ---- BAM input without pysam / samtools --------------------------------------------------------------
 * Replaces, at the reference's pysam boundary, `pysam.AlignmentFile(bam).fetch(chr)` with its per-record fields
 * (extract_reads_signature.py:68-105, 160-209: reference_name, pos, reference_end, cigar, qname, is_reverse, mapq, flag) and the
 * `samtools view bam chr:start-end` crop of 1_crop_bam.py:74.  BGZF blocks are inflated with zlib on the host; a .bai next to the
 * file (x.bam.bai or x.bai) gives the start of a region, without one the file is scanned from its first record.  No GPU needed
 * for fsv_bam_*; fsv_read_signatures runs the CIGAR scan (extract_sig_from_cigar, extract_reads_signature.py:11-44) as a kernel. */
typedef struct fsv_bam fsv_bam;
int  fsv_bam_open(const char *path, fsv_bam **out);
void fsv_bam_close(fsv_bam *bam);
int  fsv_bam_n_refs(const fsv_bam *bam);
const char *fsv_bam_ref_name(const fsv_bam *bam, int ref_id);
int64_t fsv_bam_ref_length(const fsv_bam *bam, int ref_id);   /* -1: no such reference sequence (pysam get_reference_length) */
int  fsv_bam_ref_id(const fsv_bam *bam, const char *name);   /* -1: no such reference sequence */
int  fsv_bam_has_index(const fsv_bam *bam);
void fsv_bam_set_threads(fsv_bam *bam, int n);   /* host threads inflating BGZF blocks (default: the machine's, at most 16) */

typedef struct fsv_bam_records {
    /* all host; caller-allocated with the capacities below (a first call with pos == NULL only counts) */
    int32_t  *pos, *ref_end;      /* 0-based start, exclusive end on the reference (pysam pos / reference_end) */
    uint16_t *flag;
    uint8_t  *mapq;
    uint64_t *cigar_off;          /* record r owns cigar[cigar_off[r] .. + n_cigar_op[r]) */
    uint32_t *n_cigar_op;
    uint64_t *qname_off;          /* NUL-terminated name of record r at qname + qname_off[r] */
    int32_t  *l_seq;
    uint32_t *cigar;              /* BAM encoding len << 4 | op */
    char     *qname;
    uint64_t *seq_word_off;       /* want_seq: bases of record r, 2 bits each as in the read store, from seq_words_buf[seq_word_off[r]] */
    uint32_t *seq_words_buf;
    char     *seq_ascii;          /* want_seq & 2: bases of record r as text (pysam read.seq), l_seq[r] bytes from seq_ascii + seq_ascii_off[r] */
    uint64_t *seq_ascii_off;
    int32_t  *ref_id;             /* optional: reference sequence of the record (-1 unmapped) */
    char     *sa;                 /* want_seq & 4: NUL-terminated text of the record's SA tag ("" when it has none) at sa + sa_off[r] */
    uint64_t *sa_off;
    int32_t  *ps, *hp;            /* optional (both or neither): integer PS / HP tags of the record, FSV_BAM_NO_TAG when absent (output_fas.py:31-33) */
    uint64_t rec_cap, cigar_cap, qname_cap, seq_cap, seq_ascii_cap, sa_cap;
    uint64_t n_rec, n_cigar, qname_bytes, seq_words, seq_ascii_bytes, sa_bytes;   /* out */
} fsv_bam_records;
#define FSV_BAM_NO_TAG (-2147483647 - 1)
/* mapped records of reference ref_id that overlap [beg, end) (end <= 0: to the end), in file order; ref_id -1: every record of the
 * file (pysam fetch(until_eof=True), output_fas.py:26).  want_seq: bit 0 the 2-bit words, bit 1 the text, bit 2 the SA tags (supplementary alignments, Reads_Based_Scan.py:527-531) */
int fsv_bam_fetch(fsv_bam *bam, int ref_id, int64_t beg, int64_t end, fsv_bam_records *out, int want_seq);

typedef struct fsv_read_sig {
    uint32_t rec;                 /* index into the fetched records */
    uint32_t type;                /* 0 DEL, 1 INS */
    int32_t  ref_pos, len;        /* 0-based reference position of the event, its length */
    int32_t  read_off;            /* offset in the read, hard clip at the head included (extract_reads_signature.py:19-24) */
    uint32_t pad;
} fsv_read_sig;                   /* 24 bytes */
/* DEL / INS of at least min_svlen in the CIGARs of records with mapq >= min_mapq; order of `out` is unspecified (sort by rec, read_off) */
int fsv_read_signatures(fsv_ctx *ctx, const fsv_bam_records *rec, int min_mapq, int min_svlen, fsv_read_sig *out, uint32_t cap, uint32_t *n_out);

#ifdef __cplusplus
}
#endif
#endif /* FOCALSV_HIP_H */
