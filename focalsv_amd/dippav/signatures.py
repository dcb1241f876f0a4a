"""DEL/INS signatures from contig-vs-reference alignments.

Follows focalsv/4_sv_calling/Dippav/extract_contig_signature_{CCS,CLR,ONT}.py:
  extract_sig_from_cigar   CCS:14-127   CIGAR walk (>= 30 bp) + per-contig merge of neighbouring events
  extract_sig_from_split   CCS:268-327  CLR:287-343  ONT:273-347   INS/DEL from consecutive records of one contig
  cluster_del/cluster_ins  CCS:157-249  seed-based (not transitive) clustering, longest member kept
  merge_all                CCS:434-455  cigar U split, clustered once more
  pair_sig                 CCS:504-559  hp1/hp2 pairing -> GT 1/1 or 0/1

A signature is the reference's 10-field list:
  [chrom, 'DEL'|'INS', ref_pos0, length, contig, contig_start, contig_end, '+'|'-', 'cigar'|'split-alignment', mapq|"m1-m2"]
and pair_sig appends [GT, TIG_REGION, strands, sources, mapqs].
Records are duck-typed like pysam.AlignedSegment; AlignedSegment below is what the aligner boundary produces.
"""
from collections import Counter
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
from bisect import bisect_left

BAM_M, BAM_I, BAM_D, BAM_S, BAM_H = 0, 1, 2, 4, 5


@dataclass
class AlignedSegment:
    """The eight pysam attributes the reference reads (CCS:14-47, 279-282, 347-356)."""
    reference_name: str
    pos: int                      # 0-based leftmost reference coordinate
    reference_end: int            # exclusive
    cigar: List[Tuple[int, int]]  # (op, len) with BAM op codes
    qname: str
    is_reverse: bool
    mapq: int
    seq: Optional[str] = None


@dataclass(frozen=True)
class Profile:
    """Per data type differences (CCS vs CLR vs ONT)."""
    name: str
    split_rule: str                # 'absolute' (CCS) | 'ratio'
    ratio_del: float = 0.0         # r in CLR:326-333 / ONT:315-322
    ratio_ins_lo: float = 0.0      # lower bound factor on Diffdis for INS (CLR r, ONT 0.8)
    ratio_ins_hi: float = 0.0
    contig_filter: bool = False    # CLR:384-386: ins_pct <= 0.13 or mean M-run >= 200


PROFILES = {
    "CCS": Profile("CCS", "absolute"),
    "CLR": Profile("CLR", "ratio", 0.3, 0.3, 0.3, True),
    "ONT": Profile("ONT", "ratio", 0.5, 0.8, 0.5, False),
}


def sort_sig(sig_list):
    """position order through numpy's default argsort, exactly as the reference (CCS:130-139)"""
    order = np.argsort([s[2] for s in sig_list])
    return [sig_list[i] for i in order]


# ---------------------------------------------------------------------------------------------- CIGAR
def _merge_neighbours(events, can_merge, merge):
    if len(events) < 2:
        return events
    out = [events[0]]
    for nxt in events[1:]:
        if can_merge(out[-1], nxt):
            out[-1] = merge(out[-1], nxt)
        else:
            out.append(nxt)
    return out


def extract_sig_from_cigar(read, min_svlen=30):
    """-> (del_sigs, ins_sigs, ref_end, contig_len_seen).  Ops other than M/I/D/S are skipped except that a
    leading hard clip shifts contig offsets (CCS:24-26, 41-45)."""
    chrom, qname, mapq = read.reference_name, read.qname, read.mapq
    strand = '-' if read.is_reverse else '+'
    ref, ctg = read.pos, 0
    head = read.cigar[0][1] if read.cigar[0][0] == BAM_H else 0
    dels, inss = [], []
    for op, n in read.cigar:
        if op == BAM_M:
            ref += n
            ctg += n
        elif op == BAM_S:
            ctg += n
        elif op == BAM_D:
            if n >= min_svlen:
                dels.append([chrom, 'DEL', ref, n, qname, ctg + head, ctg + head + 1, strand, 'cigar', mapq])
            ref += n
        elif op == BAM_I:
            if n >= min_svlen:
                inss.append([chrom, 'INS', ref, n, qname, ctg + head, ctg + head + n, strand, 'cigar', mapq])
            ctg += n

    def ins_close(a, b):
        d = abs(b[2] - a[2])
        return (a[3] > 250 and b[3] > 250 and d < 250) or (a[3] > 320 and b[3] > 320 and d < 380) or (a[3] > 100 and b[3] > 100 and d < 250)

    def ins_join(a, b):
        return [a[0], a[1], a[2], b[6] - a[5], a[4], a[5], b[6], a[7], 'cigar', mapq]

    def del_close(a, b):
        return a[3] > 150 and b[3] > 150 and abs(b[2] - a[2]) < 150

    def del_join(a, b):
        return [a[0], a[1], a[2], b[2] + b[3] - a[2], a[4], a[5], a[5] + 1, a[7], 'cigar', mapq]

    return _merge_neighbours(dels, del_close, del_join), _merge_neighbours(inss, ins_close, ins_join), ref, ctg


def contig_passes_filter(read, profile: Profile):
    """CLR only (CLR:12-30, 384-386): drop contigs that are mostly insertions with short match runs."""
    if not profile.contig_filter:
        return True
    m = [n for op, n in read.cigar if op == BAM_M]
    ins = sum(n for op, n in read.cigar if op == BAM_I)
    ins_pct = ins / (sum(m) + ins)
    return ins_pct <= 0.13 or (sum(m) / len(m)) >= 200


# ---------------------------------------------------------------------------------------------- split alignments
def _query_len(cigar):
    return sum(n for op, n in cigar if op in (BAM_M, BAM_I, BAM_S, BAM_H))


def get_read_start_end(cigar):
    """aligned interval on the contig (CCS:251-266)"""
    rl = _query_len(cigar)
    start = cigar[0][1] if cigar[0][0] in (BAM_S, BAM_H) else 0
    end = rl - cigar[-1][1] if cigar[-1][0] in (BAM_S, BAM_H) else rl
    return start, end


def extract_sig_from_split(read1, read2, min_mapq=50, max_svlen=50000, profile: Profile = PROFILES["CCS"]):
    """two consecutive records (BAM order) of one contig -> (del_sigs, ins_sigs)"""
    assert read1.pos <= read2.pos and read1.qname == read2.qname and read1.reference_name == read2.reference_name
    dels, inss = [], []
    c1, c2 = read1.cigar, read2.cigar
    if not (read1.is_reverse == read2.is_reverse and read1.mapq >= min_mapq and read2.mapq >= min_mapq
            and c1[-1][0] in (BAM_S, BAM_H) and c2[0][0] in (BAM_S, BAM_H)):
        return dels, inss
    rl1, rl2 = _query_len(c1), _query_len(c2)
    assert rl1 == rl2
    ref1e, ref2s = read1.reference_end, read2.pos
    q1e, q2s = rl1 - c1[-1][1], c2[0][1]
    diffdis = (ref2s - ref1e) - (q2s - q1e)
    diffolp = ref1e - ref2s
    strand = '-' if read1.is_reverse else '+'
    chrom, qname = read1.reference_name, read1.qname
    mq = "%d-%d" % (read1.mapq, read2.mapq)
    if abs(diffdis) > max_svlen:
        return dels, inss

    def ins_sig():
        svlen = abs(q2s - q1e + diffolp)
        pos = int((ref1e + ref2s) / 2) if abs(diffolp) > 400 else ref2s
        return [chrom, 'INS', pos, svlen, qname, q1e - diffolp, q2s, strand, 'split-alignment', mq]

    if profile.split_rule == 'absolute':
        if diffolp < 30 and diffdis >= 30:
            dels.append([chrom, 'DEL', ref1e, diffdis, qname, q1e, q2s, strand, 'split-alignment', mq])
        elif diffolp < 3000 and diffdis >= 30:
            dels.append([chrom, 'DEL', ref1e - diffdis, diffdis, qname, q1e - diffdis, q2s - diffdis, strand, 'split-alignment', mq])
        elif diffolp < 3000 and diffdis <= -30:
            inss.append(ins_sig())
    else:
        if diffdis >= 30:
            qolp = q1e - q2s  # the ratio rules look at the overlap on the contig for deletions (CLR:330)
            if -(diffdis * profile.ratio_del) <= qolp <= diffdis * profile.ratio_del:
                dels.append([chrom, 'DEL', ref1e, diffdis, qname, q1e, q2s, strand, 'split-alignment', mq])
        elif diffdis * profile.ratio_ins_lo <= diffolp <= abs(diffdis) * profile.ratio_ins_hi and diffdis <= -30:
            inss.append(ins_sig())
    return dels, inss


# ---------------------------------------------------------------------------------------------- clustering
def _seed_cluster(sigs, same, max_shift=None):
    """the first unassigned signature seeds a cluster and absorbs every later unassigned one that matches it;
    the longest member represents the cluster (first on ties).  CCS:157-249.
    The reference scans all pairs; a match needs |dpos| <= max_shift, so on position-sorted input (which is what
    the reference always passes) the scan can stop once positions run past the seed -- same result, O(n) not O(n^2)."""
    label = [-1] * len(sigs)
    is_sorted = max_shift is not None and all(sigs[k][2] <= sigs[k + 1][2] for k in range(len(sigs) - 1))
    for i, a in enumerate(sigs):
        if label[i] != -1:
            continue
        label[i] = i
        for j in range(i + 1 if is_sorted else 0, len(sigs)):
            b = sigs[j]
            if is_sorted and b[2] - a[2] > max_shift:
                break
            if label[j] == -1 and same(a, b):
                label[j] = i
    # one representative per cluster, clusters in first-seen order (the reference iterates a Counter of the labels), the
    # longest member, first on ties
    best = {}
    for s, l in zip(sigs, label):
        b = best.get(l)
        if b is None or s[3] > b[3]:
            best[l] = s
    return list(best.values())


def _del_match(a, b, max_shift, min_overlap, min_size_sim):
    s1, s2 = a[2], b[2]
    e1, e2 = s1 + a[3], s2 + b[3]
    overlap = (min(e1, e2) - max(s1, s2)) / min(a[3], b[3])
    return abs(s1 - s2) <= max_shift and overlap >= min_overlap and min(a[3], b[3]) / max(a[3], b[3]) >= min_size_sim


def _ins_match(a, b, max_shift, min_size_sim):
    return abs(a[2] - b[2]) <= max_shift and min(a[3], b[3]) / max(a[3], b[3]) >= min_size_sim


def cluster_del(sigs, max_shift=100, min_overlap_ratio=0.5, min_size_similarity=0.5):
    return _seed_cluster(sigs, lambda a, b: _del_match(a, b, max_shift, min_overlap_ratio, min_size_similarity), max_shift)


def cluster_ins(sigs, max_shift=100, min_size_similarity=0.5):
    return _seed_cluster(sigs, lambda a, b: _ins_match(a, b, max_shift, min_size_similarity), max_shift)


def merge_all(del_cigar, ins_cigar, del_split, ins_split):
    ins_final = cluster_ins(sort_sig(ins_cigar + ins_split))
    del_final = cluster_del(sort_sig(del_cigar + del_split))
    return sort_sig(ins_final + del_final)


def signatures_one_hap(records: Sequence, hp: str, profile: Profile = PROFILES["CCS"], min_cigar_mapq=50, min_split_mapq=50):
    """extract_signature_one_hap (CCS:457-469) over records in BAM order (sorted by reference position).
    `hp in qname` selects the haplotype exactly as the reference does (CCS:348)."""
    dels, inss = [], []
    for r in records:
        if hp in r.qname and r.mapq >= min_cigar_mapq and contig_passes_filter(r, profile):
            d, i, ref_end, ctg = extract_sig_from_cigar(r, 30)
            assert ref_end == r.reference_end
            if r.seq:
                assert len(r.seq) == ctg
            dels += d
            inss += i
    dels_c, inss_c = cluster_del(sort_sig(dels)), cluster_ins(sort_sig(inss))
    names = [r.qname for r in records if hp in r.qname and r.mapq >= min_split_mapq]
    multi = {n for n, c in Counter(names).items() if c > 1}
    kept = [r for r in records if r.qname in multi and r.mapq >= min_split_mapq]
    sd, si = [], []
    for name in [n for n in Counter(names) if n in multi]:
        mine = [r for r in kept if r.qname == name]
        for a, b in zip(mine[:-1], mine[1:]):
            d, i = extract_sig_from_split(a, b, min_split_mapq, 50000, profile)
            sd += d
            si += i
    sd_c, si_c = cluster_del(sort_sig(sd)), cluster_ins(sort_sig(si))
    return merge_all(dels_c, inss_c, sd_c, si_c)


# ---------------------------------------------------------------------------------------------- haplotype pairing
def pair_sig(sig_hp1, sig_hp2, max_compare_dist=1000, max_shift=200, min_overlap_ratio=0.5, min_size_similarity=0.5):
    """CCS:504-559.  Note the reference ignores its max_shift/ratio arguments and hard-codes 200 / 0.5 / 0.5."""
    partner1, partner2 = [-1] * len(sig_hp1), [-1] * len(sig_hp2)
    pos2 = [b[2] for b in sig_hp2]
    in_order = all(pos2[k] <= pos2[k + 1] for k in range(len(pos2) - 1))   # the reference always passes sort_sig output
    for i, a in enumerate(sig_hp1):
        # a pair needs |dpos| <= 200: signatures more than that to the left cannot match, so the scan may start behind them
        j0 = bisect_left(pos2, a[2] - 200) if in_order else 0
        for j in range(j0, len(sig_hp2)):
            b = sig_hp2[j]
            if b[2] - a[2] > max_compare_dist:
                break
            if a[:2] == b[:2] and partner2[j] == -1:
                ok = _del_match(a, b, 200, 0.5, 0.5) if a[1] == 'DEL' else _ins_match(a, b, 200, 0.5)
                if ok:
                    partner1[i], partner2[j] = j, i
                    break
    out = []

    def tig(s):
        return "%s:%d-%d" % (s[4], s[5], s[6])

    for i, a in enumerate(sig_hp1):
        if partner1[i] == -1:
            out.append(a + ['0/1', tig(a), a[7], a[8], str(a[9])])
        else:
            b = sig_hp2[partner1[i]]
            extra = ['1/1', tig(a) + ',' + tig(b), a[7] + ',' + b[7], a[8] + ',' + b[8], str(a[9]) + ',' + str(b[9])]
            out.append((a if a[3] > b[3] else b) + extra)
    for j, b in enumerate(sig_hp2):
        if partner2[j] == -1:
            out.append(b + ['0/1', tig(b), b[7], b[8], str(b[9])])
    return sort_sig(out)
