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


def reads_signatures(records, min_mapq=50):
    """all reads of one chromosome -> position-sorted signature list (cigar source only; the split source of
    extract_reads_signature.py:108-209 needs supplementary records, which region-cropped inputs do not carry)"""
    sigs = []
    for r in records:
        if r.mapq >= min_mapq:
            d, i = extract_sig_from_cigar(r, 30)
            sigs += d + i
    order = np.argsort([s[2] for s in sigs])
    return [sigs[k] for k in order]


def write_reads_sig(sigs, output_dir, chrom):
    d = os.path.join(output_dir, "reads_signature")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "%s_reads_sig.txt" % chrom)
    with open(path, "w") as f:
        for s in sigs:
            f.write('\t'.join(str(x) for x in s) + '\n')
    return path
