"""Read-level signatures feeding the FP filter (focalsv/4_sv_calling/Dippav/extract_reads_signature.py:11-44,
108-158, 228-261): the same CIGAR walk as for contigs but without per-read merging and with 8-field records
[chrom, type, ref_pos0, length, read, read_offset, strand, 'cigar'|'split-alignment'].  Written as
reads_signature/chr<N>_reads_sig.txt: the first four fields are what FP_filter_v1.load_sig reads back."""
import os

import numpy as np

from .signatures import BAM_D, BAM_H, BAM_I, BAM_M, BAM_S


def extract_sig_from_cigar(read, min_svlen=30):
    chrom, qname = read.reference_name, read.qname
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
                dels.append([chrom, 'DEL', ref, n, qname, ctg + head, strand, 'cigar'])
            ref += n
        elif op == BAM_I:
            if n >= min_svlen:
                inss.append([chrom, 'INS', ref, n, qname, ctg + head, strand, 'cigar'])
            ctg += n
    return dels, inss


def _sort(sigs):
    order = np.argsort([s[2] for s in sigs])   # numpy's default argsort on the positions, as the reference (:44-54)
    return [sigs[k] for k in order]


def _read_len(cigar):
    return sum(n for op, n in cigar if op in (BAM_M, BAM_I, BAM_S, BAM_H))


def extract_sig_from_split(read1, read2, min_mapq=0, max_svlen=50000):
    """two consecutive records of one read -> (del_sigs, ins_sigs); extract_reads_signature.py:108-158 (the reference's
    spelling of the source tag is kept: it ends up in chr<N>_reads_sig.txt)"""
    assert read1.pos <= read2.pos and read1.qname == read2.qname and read1.reference_name == read2.reference_name
    dels, inss = [], []
    c1, c2 = read1.cigar, read2.cigar
    if read1.is_reverse == read2.is_reverse and read1.mapq >= min_mapq and read2.mapq >= min_mapq \
            and c1[-1][0] in (BAM_S, BAM_H) and c2[0][0] in (BAM_S, BAM_H):
        rl1, rl2 = _read_len(c1), _read_len(c2)
        assert rl1 == rl2
        ref1e, ref2s = read1.reference_end, read2.pos
        r1e, r2s = rl1 - c1[-1][1], c2[0][1]
        diffdis = (ref2s - ref1e) - (r2s - r1e)
        diffolp = ref1e - ref2s
        strand = '-' if read1.is_reverse else '+'
        if abs(diffdis) <= max_svlen:
            if diffolp < 30 and diffdis >= 30:
                dels.append([read1.reference_name, 'DEL', ref1e, diffdis, read1.qname, r1e, r2s, strand, 'split-alignemnt'])
            elif diffolp < 30 and diffdis <= -30:
                inss.append([read1.reference_name, 'INS', int((ref1e + ref2s) / 2), abs(diffdis), read1.qname, r1e, r2s, strand, 'split_alignment'])
    return dels, inss


def reads_signatures(records, min_mapq=50):
    """all records of one chromosome, in BAM (position) order -> the signature list of chr<N>_reads_sig.txt:
    CIGAR source from records with mapq >= 50 (extract_signature_from_cigar, :68-105), split source from consecutive records
    of reads that have several, any mapq (extract_sig_from_split_reads, :160-209), merged by position (merge_all, :221)."""
    dc, ic = [], []
    by_name = {}
    for r in records:
        by_name.setdefault(r.qname, []).append(r)
        if r.mapq >= min_mapq:
            d, i = extract_sig_from_cigar(r, 30)
            dc += d
            ic += i
    ds, is_ = [], []
    for name, rs in by_name.items():   # first-seen order of the names, as the reference's Counter
        for k in range(len(rs) - 1):
            d, i = extract_sig_from_split(rs[k], rs[k + 1], 0, 50000)
            ds += d
            is_ += i
    return _sort(_sort(dc) + _sort(ic) + _sort(ds) + _sort(is_))


def records_from_bam(bam_path, chrom):
    """the records pysam's fetch(chrom) yields, through the library's own BAM reader (focalsv_amd.bam; no pysam, no GPU)"""
    from .. import bam
    with bam.BamFile(bam_path) as f:
        if chrom not in f.references:
            return []
        recs = f.fetch(chrom)
    return [recs.segment(r) for r in range(len(recs))]


def write_reads_sig(sigs, output_dir, chrom):
    d = os.path.join(output_dir, "reads_signature")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "%s_reads_sig.txt" % chrom)
    with open(path, "w") as f:
        for s in sigs:
            f.write('\t'.join(str(x) for x in s) + '\n')
    return path
