"""VCF records from paired signatures (extract_contig_signature_CCS.py:582-670) and the reference's header
(focalsv/4_sv_calling/Dippav/header, 11 lines)."""

HEADER_LINES = [
    '##fileformat=VCFv4.2\n',
    '##FILTER=<ID=PASS,Description="All filters passed">\n',
    '##INFO=<ID=SVTYPE,Number=1,Type=String,Description="Type of SV:DEL=Deletion, TRA=Translocation, INS=Insertion, DUP=Duplication, INV=Inversion">\n',
    '##INFO=<ID=SVLEN,Number=.,Type=Integer,Description="Difference in length between REF and ALT alleles">\n',
    '##INFO=<ID=TIG_REGION,Number=.,Type=String,Description="Contig region where variant was found (one per alt with h1 before h2 for homozygous calls)">\n',
    '##INFO=<ID=QUERY_STRAND,Number=.,Type=String,Description="Strand of variant in the contig relative to the reference (order follows TIG_REGION)">\n',
    '##INFO=<ID=SIG_SOURCE,Number=.,Type=String,Description="Source of the variant call signature (order follows TIG_REGION)">\n',
    '##INFO=<ID=TIG_MAPQ,Number=.,Type=String,Description="Mapping quality of the contigs (order follows TIG_REGION)">\n',
    '##INFO=<ID=CollapseId,Number=1,Type=Integer,Description="collapse match ID">\n',
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n',
    '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n',
]

_COMP = str.maketrans('NATGC', 'NTACG')


def reverse_complement(seq):
    """upper-cased reverse complement; any other letter raises, like the reference's dict lookup (CCS.py:582-588)"""
    s = seq.upper()
    if s.strip('NATGC'):
        raise KeyError(s.strip('NATGC')[0])
    return s[::-1].translate(_COMP)


def allele_sequence(sig, ref_seq, contigs):
    """INS: the inserted contig bases (reverse strand: revcomp of contig[-end:-start], including the reference's
    `-0` quirk when contig_start == 0, CCS:622-625); DEL: the deleted reference bases."""
    if sig[1] == 'INS':
        c = contigs[sig[4]]
        return reverse_complement(c[-sig[6]:-sig[5]]) if sig[7] == '-' else c[sig[5]:sig[6]]
    return ref_seq[sig[2]:sig[2] + sig[3]]


def vcf_lines(paired_sigs, ref_seq, contigs):
    """body lines in signature order; signatures whose contig is unknown are dropped (CCS:603)."""
    lines, n_ins, n_del = [], 0, 0
    for sig in paired_sigs:
        if sig[4] not in contigs:
            continue
        seq = allele_sequence(sig, ref_seq, contigs)
        chrom, svtype, pos1 = sig[0], sig[1], sig[2]
        anchor = ref_seq[pos1 - 1]
        if svtype == 'DEL':
            ref_allele, alt_allele = anchor + seq, anchor
            n_del += 1
            idx = n_del
        else:
            ref_allele, alt_allele = anchor, anchor + seq
            n_ins += 1
            idx = n_ins
        info = "SVLEN=%d;SVTYPE=%s;TIG_REGION=%s;QUERY_STRAND=%s;SIG_SOURCE=%s;TIG_MAPQ=%s" % (
            len(alt_allele) - len(ref_allele), svtype, sig[11], sig[12], sig[13], sig[14])
        lines.append("%s\t%d\tdippav.%s.%s.%d\t%s\t%s\t%d\tPASS\t%s\tGT\t%s\n" % (
            chrom, pos1, chrom, svtype, idx, ref_allele.upper(), alt_allele.upper(), 20, info, sig[10]))
    return lines


def write_vcf(paired_sigs, vcf_path, ref_seq, contigs, header_lines=None):
    with open(vcf_path, 'w') as f:
        f.writelines(header_lines or HEADER_LINES)
        f.writelines(vcf_lines(paired_sigs, ref_seq, contigs))
