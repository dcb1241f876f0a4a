"""Read-based draft calls for the CLR / ONT post-processing branch: the clustering / genotyping half of
focalsv/5_post_processing/Reads_Based_Scan (a cuteSV derivative) -- resolveINDEL.py (resolution_DEL / resolution_INS :18-99, :193-279,
generate_del_cluster / generate_ins_cluster :101-191, :280-376, call_gt :384-417), genotype.py (cal_GL, cal_CIPOS,
threshold_ref_count, count_coverage :10-85, the DEL / INS records of generate_output :87-143) and the collection order of
Reads_Based_Scan.py's main_ctrl (:683-741) -- on DEL.sigs / INS.sigs (focalsv_amd.reads_scan) and the library's own BAM reader.
Output: reads_draft_variants.vcf, what gt_impute / final_process_ont take their genotypes and insertions from."""
import os
import time
from math import log10
from typing import Dict, List

import numpy as np

from . import bam as B, fasta

# FocalSV_Filter_GT_Correct.py:118-134: clustering parameters per data type (max_cluster_bias_INS, diff_ratio_merging_INS, .._DEL, .._DEL)
CLUSTER_PARA = {'Hifi': (1000, 0.9, 1000, 0.5), 'CLR': (100, 0.3, 200, 0.5), 'ONT': (100, 0.3, 100, 0.3)}
MIN_SUPPORT, GT_ROUND = 10, 500            # Description.py defaults: --min_support, --gt_round
_ERR, _PRIOR = 0.1, float(1 / 3)
_GENOTYPES = ["0/0", "0/1", "1/1"]


# ---------------------------------------------------------------------------------------------------- genotype.py
def _cal_gl(c0, c1):
    """genotype, PL, GQ, QUAL from reference-supporting (c0) and variant-supporting (c1) read counts"""
    total = c0 + c1
    if total > 100:
        c0 = int(100 * float(c0 / total))
        c1 = 100 - c0
    gl00 = np.float64(pow((1 - _ERR), c0) * pow(_ERR, c1) * (1 - _PRIOR) / 2)
    gl11 = np.float64(pow(_ERR, c0) * pow((1 - _ERR), c1) * (1 - _PRIOR) / 2)
    gl01 = np.float64(pow(0.5, c0 + c1) * _PRIOR)
    lp = np.array([log10(gl00), log10(gl01), log10(gl11)])
    m = max(lp)
    lse = m + log10(sum(pow(10.0, x - m) for x in lp))
    prob = list(np.minimum(lp - lse, 0.0))
    p = [pow(10, i) for i in prob]
    pl = [int(np.around(-10 * log10(i))) for i in p]
    gq = [int(-10 * log10(p[1] + p[2])), int(-10 * log10(p[0] + p[2])), int(-10 * log10(p[0] + p[1]))]
    qual = abs(np.around(-10 * log10(p[0]), 1))
    return _GENOTYPES[prob.index(max(prob))], "%d,%d,%d" % (pl[0], pl[1], pl[2]), max(gq), qual


def _cipos(std, num):
    pos = int(1.96 * std / num ** 0.5)
    return "-%d,%d" % (pos, pos)


def _ref_count_bound(num):
    if num <= 2:
        return 10 * num
    if num <= 5:
        return 5 * num
    if num <= 15:
        return 4 * num
    return 3 * num


class ChromReads:
    """the records of one chromosome as arrays, for the repeated fetch(chr, s, e) of the genotyping step"""

    def __init__(self, bamfile):
        self._bam = B.BamFile(bamfile)
        self._chrom: Dict[str, tuple] = {}
        self.ref_len = {}

    def close(self):
        self._bam.close()

    def _load(self, chrom):
        if chrom not in self._chrom:
            r = self._bam.fetch(chrom)
            pos, end = r.pos.astype(np.int64), r.ref_end.astype(np.int64)
            self._chrom[chrom] = (pos, end, r.flag.astype(np.int64), r.names, int((end - pos).max()) if len(pos) else 0)
        return self._chrom[chrom]

    def spanning_reads(self, chrom, s, e, up_bound, itround):
        """count_coverage (genotype.py:64-85): names of primary reads (flag 0 / 16) that start before s and end after e, among the
        records overlapping [s, e) in file order; gives up after itround records -> (status, names)"""
        pos, end, flag, names, longest = self._load(chrom)
        a = int(np.searchsorted(pos, s - longest, 'left'))
        b = int(np.searchsorted(pos, e, 'left'))
        idx = np.nonzero(end[a:b] > s)[0] + a
        got, iteration, primary = set(), 0, 0
        for i in idx:
            iteration += 1
            if flag[i] in (0, 16):
                primary += 1
                if pos[i] < s and end[i] > e:
                    got.add(names[i])
                    if len(got) >= up_bound:
                        return 1, got
            else:
                continue
            if iteration >= itround:
                return (1 if float(primary / iteration) <= 0.2 else -1), got
        return 0, got


def _call_gt(reads: ChromReads, ref_len, search_pos, chrom, read_ids, max_cluster_bias, gt_round):
    s = max(int(search_pos) - max_cluster_bias, 0)
    e = min(int(search_pos) + max_cluster_bias, ref_len)
    status, spanning = reads.spanning_reads(chrom, s, e, _ref_count_bound(len(read_ids)), gt_round)
    if status == -1:
        return len(read_ids), '.', "./.", ".,.,.", ".", "."
    dr = sum(1 for q in spanning if q not in read_ids)
    gt, gl, gq, qual = _cal_gl(dr, len(read_ids))
    return len(read_ids), dr, gt, gl, gq, qual


# ---------------------------------------------------------------------------------------------------- resolveINDEL.py
def _alleles(cluster, read_count, ratio, n_fields):
    """one position cluster -> alleles: the longest signature per read, sorted by length, cut where the length jumps by more than
    ratio x the mean length; alleles in increasing order of support"""
    by_read = {}
    for el in cluster:
        if el[2] not in by_read or el[1] > by_read[el[2]][1]:
            by_read[el[2]] = el
    if len(by_read) < read_count:
        return []
    srt = sorted(by_read.values(), key=lambda x: x[1])
    jump = ratio * np.mean([i[1] for i in srt])
    last = srt[0][1]
    alleles = [[[srt[0][0]], [srt[0][1]], [], [srt[0][2]]] + ([[srt[0][3]]] if n_fields == 4 else [])]
    for i in srt[1:]:
        if i[1] - last > jump:
            alleles[-1][2].append(len(alleles[-1][0]))
            alleles.append([[], [], [], []] + ([[]] if n_fields == 4 else []))
        alleles[-1][0].append(i[0])
        alleles[-1][1].append(i[1])
        alleles[-1][3].append(i[2])
        if n_fields == 4:
            alleles[-1][4].append(i[3])
        last = i[1]
    alleles[-1][2].append(len(alleles[-1][0]))
    return sorted(alleles, key=lambda x: x[2])


def _resolve(sig_path, chrom, svtype, read_count, ratio, max_cluster_bias, min_support, reads, ref_len, genotype, gt_round):
    """resolution_DEL / resolution_INS: signatures of one chromosome in file (position) order, a new cluster when the next one is
    more than max_cluster_bias past the previous"""
    ins = svtype == 'INS'
    out = []

    def emit(cluster):
        if len(cluster) < read_count or (cluster[-1][0] == 0 and cluster[-1][1] == 0):
            return
        for al in _alleles(cluster, read_count, ratio, 4 if ins else 3):
            if al[2][0] < min_support:
                continue
            start = np.mean(al[0])
            cip = _cipos(np.std(al[0]), len(al[0]))
            length = np.mean(al[1])
            cil = _cipos(np.std(al[1]), len(al[1]))
            seq = None
            if ins:
                seq = next((s[0:int(length)] for s in al[4] if len(s) >= int(length)), None)
                if seq is None:
                    continue
            if genotype:
                where, bias = (int(start), 1000) if ins else (int(np.min(al[0])), max_cluster_bias)
                _, dr, gt, gl, gq, qual = _call_gt(reads, ref_len, where, chrom, al[3], bias, gt_round)
            else:
                dr, gt, gl, gq, qual = '.', './.', '.,.,.', '.', '.'
            rec = [chrom, svtype, str(int(start)), str(int(length) if ins else int(-length)), str(al[2][0]), str(cip), str(cil), str(dr), str(gt),
                   str(gl), str(gq), str(qual), str(','.join(al[3]))]
            out.append(rec + [seq] if ins else rec)

    cluster = [[0, 0, '', ''] if ins else [0, 0, '']]
    with open(sig_path) as f:
        for line in f:
            seq = line.strip('\n').split('\t')
            if seq[1] != chrom:
                continue
            el = [int(seq[2]), int(seq[3]), seq[4]] + ([seq[5] if len(seq) > 5 else ''] if ins else [])
            if el[0] - cluster[-1][0] > max_cluster_bias:
                emit(cluster)
                cluster = [el]
            else:
                cluster.append(el)
    emit(cluster)
    return out


# ---------------------------------------------------------------------------------------------------- output
def _header(contigs, sample, argv_text):
    h = ["##fileformat=VCFv4.2\n", "##source=cuteSV-1.0.11\n", "##fileDate=%s\n" % time.strftime('%Y-%m-%d %H:%M:%S %w-%Z', time.localtime())]
    h += ["##contig=<ID=%s,length=%d>\n" % (c, n) for c, n in contigs]
    for k, d in (("INS", "Insertion of novel sequence relative to the reference"), ("DEL", "Deletion relative to the reference"),
                 ("DUP", "Region of elevated copy number relative to the reference"), ("INV", "Inversion of reference sequence"),
                 ("BND", "Breakend of translocation")):
        h.append('##ALT=<ID=%s,Description="%s">\n' % (k, d))
    for k, num, typ, d in (("PRECISE", "0", "Flag", "Precise structural variant"), ("IMPRECISE", "0", "Flag", "Imprecise structural variant"),
                           ("SVTYPE", "1", "String", "Type of structural variant"),
                           ("SVLEN", "1", "Integer", "Difference in length between REF and ALT alleles"),
                           ("CHR2", "1", "String", "Chromosome for END coordinate in case of a translocation"),
                           ("END", "1", "Integer", "End position of the variant described in this record"),
                           ("CIPOS", "2", "Integer", "Confidence interval around POS for imprecise variants"),
                           ("CILEN", "2", "Integer", "Confidence interval around inserted/deleted material between breakends"),
                           ("RE", "1", "Integer", "Number of read support this record"),
                           ("STRAND", "A", "String", "Strand orientation of the adjacency in BEDPE format (DEL:+-, DUP:-+, INV:++/--)"),
                           ("RNAMES", ".", "String", "Supporting read names of SVs (comma separated)")):
        h.append('##INFO=<ID=%s,Number=%s,Type=%s,Description="%s">\n' % (k, num, typ, d))
    h.append('##FILTER=<ID=q5,Description="Quality below 5">\n')
    for k, num, typ, d in (("GT", "1", "String", "Genotype"), ("DR", "1", "Integer", "# High-quality reference reads"),
                           ("DV", "1", "Integer", "# High-quality variant reads"),
                           ("PL", "G", "Integer", "# Phred-scaled genotype likelihoods rounded to the closest integer"),
                           ("GQ", "1", "Integer", "# Genotype quality")):
        h.append('##FORMAT=<ID=%s,Number=%s,Type=%s,Description="%s">\n' % (k, num, typ, d))
    h.append('##CommandLine="cuteSV %s"\n' % argv_text)
    h.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s\n" % sample)
    return h


def vcf_records(results, ref_seq: Dict[str, str], report_readid=False) -> List[str]:
    """the DEL / INS lines of generate_output (genotype.py:100-143)"""
    lines, svid = [], {"INS": 0, "DEL": 0}
    for i in results:
        pos, ln = int(i[2]), i[3]
        ref = ref_seq[i[0]]
        end = pos + 1 if i[1] == "INS" else pos + 1 + abs(int(float(ln)))
        info = "%s;SVTYPE=%s;SVLEN=%s;END=%s;CIPOS=%s;CILEN=%s;RE=%s;RNAMES=%s" % (
            "IMPRECISE" if i[8] == "0/0" else "PRECISE", i[1], ln, str(end), i[5], i[6], i[4], i[12] if report_readid else "NULL")
        if i[1] == "DEL":
            info += ";STRAND=+-"
        flt = "PASS" if i[11] == "." or i[11] is None else ("PASS" if float(i[11]) >= 5.0 else "q5")
        a0 = max(pos - 1, 0)
        if i[1] == "INS":
            ref_al, alt_al = str(ref[a0]), str(ref[a0]) + i[13]
        else:
            ref_al, alt_al = str(ref[a0:pos - int(ln)]), str(ref[a0])
        lines.append("%s\t%s\tcuteSV.%s.%d\t%s\t%s\t%s\t%s\t%s\tGT:DR:DV:PL:GQ\t%s:%s:%s:%s:%s\n" % (
            i[0], str(pos + 1), i[1], svid[i[1]], ref_al, alt_al, i[11], flt, info, i[8], i[7], i[4], i[9], i[10]))
        svid[i[1]] += 1
    return lines


def draft_calls(bamfile, reference, out_vcf, sigdir, dtype='Hifi', chromosome='wgs', genotype=True, sample="NULL", argv_text=""):
    """Reads_Based_Scan.py's main_ctrl after the signature collection: cluster DEL.sigs / INS.sigs of <sigdir> per chromosome,
    genotype every allele from the reads spanning it, write <out_vcf> (reads_draft_variants.vcf) sorted by chromosome and position"""
    bias_ins, ratio_ins, bias_del, ratio_del = CLUSTER_PARA[dtype if dtype in CLUSTER_PARA else 'ONT']
    use = ['chr%d' % i for i in range(1, 23)] if str(chromosome) == 'wgs' else ['chr%s' % chromosome]
    reads = ChromReads(bamfile)
    try:
        with B.BamFile(bamfile) as f:
            lens = dict(zip(f.references, f.reference_lengths))
        contigs = [(c, lens[c]) for c in lens if c in use]
        present = {}
        for svtype in ("DEL", "INS"):
            seen = []
            for line in open(os.path.join(sigdir, svtype + ".sigs")):
                c = line.strip('\n').split('\t')[1]
                if c not in seen:
                    seen.append(c)
            present[svtype] = sorted(seen)
        results = []
        for svtype, bias, ratio in (("DEL", bias_del, ratio_del), ("INS", bias_ins, ratio_ins)):
            for chrom in present[svtype]:
                try:
                    results += _resolve(os.path.join(sigdir, svtype + ".sigs"), chrom, svtype, MIN_SUPPORT, ratio, bias, min(MIN_SUPPORT, 5), reads,
                                        lens.get(chrom, 0), genotype, GT_ROUND)
                except (KeyError, ValueError, IndexError):
                    pass     # the reference drops a chromosome whose worker raised (`try: res.get() except: pass`)
    finally:
        reads.close()
    results = sorted(results, key=lambda x: (x[0], int(x[2])))
    ref_seq = fasta.read_fasta_dict(reference)
    with open(out_vcf, 'w') as f:
        f.writelines(_header(contigs, sample, argv_text))
        f.writelines(vcf_records(results, ref_seq))
    return out_vcf
