"""dippav_variant_call on the GPU aligner: mirrors focalsv/4_sv_calling/Dippav/DipPAV_variant_call.py:52-171.

Same function name, arguments and output files: <output_dir>/{hp1.fa, hp2.fa, assemblies.fa, dippav_variant_chr<N>.vcf,
dippav_raw_variant.vcf, dippav_variant_filtered.vcf, final_vcf/dippav_variant_no_redundancy.vcf, signature/*.txt}.
`minimap2 -a -x asm5 --cs -r2k | samtools sort` (lines 103-112) is replaced by fsv_align_batch: every contig is aligned
against the reference window of its own region, so contig FASTA headers must carry the region tag
(Region_<chr>_S<start>_E<end>, which the assembly step's headers do) or `regions=` must map contig index -> window.
"""
import logging
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .. import _lib, fasta
from . import fp_filter, reads_signature, redundancy, signatures as S, vcf

FLANK = 7000  # 0_define_region.py's flank: windows are widened by this much on both sides when possible


class WindowedRef:
    """reference bases known only inside windows, indexable like the chromosome string the reference loads"""

    def __init__(self):
        self.wins: List[Tuple[int, str]] = []
        self._starts: List[int] = []

    def add(self, start: int, seq: str):
        self.wins.append((start, seq))
        self.wins.sort()
        self._starts = [w[0] for w in self.wins]

    def _find(self, pos):
        from bisect import bisect_right
        i = bisect_right(self._starts, pos) - 1
        if i >= 0:
            s, q = self.wins[i]
            if pos < s + len(q):
                return s, q
        raise IndexError(f"reference position {pos} outside the loaded windows")

    def __getitem__(self, key):
        if isinstance(key, slice):
            a, b = key.start or 0, key.stop
            if b <= a:
                return ''
            s, q = self._find(a)
            return q[a - s:b - s]
        s, q = self._find(key)
        return q[key - s]


def records_from_alignment(rec, cigar, names: Sequence[str], chrom_of: Sequence[str], offset_of: Sequence[int]) -> List[S.AlignedSegment]:
    """fsv_aln_rec[] -> pysam-like records, sorted by (chrom, pos) like `samtools sort` output"""
    out = []
    for r in rec:
        i = int(r["contig"])
        cg = cigar[int(r["cigar_off"]): int(r["cigar_off"]) + int(r["n_cigar"])]
        ops = [(int(c) & 0xf, int(c) >> 4) for c in cg]
        for name in ([names[i]] if isinstance(names[i], str) else names[i]):   # a contig filed under both haplotypes has two names
            out.append(S.AlignedSegment(chrom_of[i], int(r["ref_start"]) + offset_of[i], int(r["ref_end"]) + offset_of[i],
                                        ops, name, bool(r["rev"]), int(r["mapq"]), None))
    order = sorted(range(len(out)), key=lambda k: (out[k].reference_name, out[k].pos))
    return [out[k] for k in order]


def call_chromosome(records: Sequence[S.AlignedSegment], chrom: str, ref_seq, contigs: Dict[str, str], data_type: str = 'CCS'):
    """extract_contig_sig_* for one chromosome (CCS.py:738-752): signatures per haplotype, pairing, VCF body lines"""
    prof = S.PROFILES[data_type]
    recs = [r for r in records if r.reference_name == chrom]
    hp1 = S.signatures_one_hap(recs, 'hp1', prof)
    hp2 = S.signatures_one_hap(recs, 'hp2', prof)
    paired = S.pair_sig(hp1, hp2, 1000, 200, 0.5, 0.5)
    return paired, vcf.vcf_lines(paired, ref_seq, contigs)


def dippav_variant_call(data_type, read_bam_file, reference_path, hp1_contig_path, hp2_contig_path, output_dir, chr_num,
                        header_file=None, n_thread=10, mem_per_thread='1G', *, regions=None, read_records=None,
                        ctx: Optional[_lib.Context] = None, device: int = 0):
    """Drop-in for DipPAV_variant_call.dippav_variant_call.  Extra keyword-only arguments:
      regions       {contig_name: (chrom, win_start, win_end)} when the FASTA headers carry no region tag
      read_records  AlignedSegment-like records of the chromosome's reads, instead of read_bam_file (which is read with the
                    library's own BAM reader, focalsv_amd.bam, and its CIGARs scanned on the GPU)
    """
    assert data_type in ('CCS', 'CLR', 'ONT')
    logger = logging.getLogger(" ")
    os.makedirs(output_dir, exist_ok=True)
    header = open(header_file).readlines() if header_file else vcf.HEADER_LINES
    chrom = 'chr%d' % chr_num
    chrom_seq = fasta.read_fasta_dict(reference_path)[chrom].upper()
    with open(os.path.join(output_dir, "ref_chr%d.fa" % chr_num), 'w') as f:
        f.write('>' + chrom + '\n' + fasta.fold(chrom_seq, 60) + '\n')
    # reformat_fasta (DipPAV_variant_call.py:14-23): contig_hp{1,2}_<i>, concatenated into assemblies.fa
    names, seqs, wins = [], [], []
    for hp, path in (('hp1', hp1_contig_path), ('hp2', hp2_contig_path)):
        with open(os.path.join(output_dir, hp + '.fa'), 'w') as fw:
            for i, (hdr, seq) in enumerate(fasta.read_fasta(path)):
                name = 'contig_%s_%d' % (hp, i)
                fw.write('>' + name + '\n' + fasta.fold(seq) + '\n')
                reg = (regions or {}).get(hdr.split()[0]) or fasta.parse_region(hdr)
                if reg is None:
                    raise ValueError(f"contig '{hdr}' of {path} carries no Region_<chr>_S<s>_E<e> tag and no regions= entry; "
                                     "the GPU aligner needs the reference window of every contig")
                names.append(name); seqs.append(seq.upper()); wins.append(reg)
    with open(os.path.join(output_dir, 'assemblies.fa'), 'w') as f:
        for hp in ('hp1', 'hp2'):
            f.write(open(os.path.join(output_dir, hp + '.fa')).read())
    # one reference window per distinct region, widened by FLANK
    win_index, refs, ref_start = {}, [], []
    for reg in wins:
        if reg not in win_index:
            a, b = max(0, reg[1] - FLANK), min(len(chrom_seq), reg[2] + FLANK)
            win_index[reg] = len(refs)
            refs.append(chrom_seq[a:b].encode()); ref_start.append(a)
    own = ctx is None
    ctx = ctx or _lib.Context(device)
    try:
        keep = [i for i, w in enumerate(wins) if w[0] == chrom and seqs[i]]
        rec, cigar, status = ctx.align_batch([seqs[i].encode() for i in keep], [win_index[wins[i]] for i in keep], refs)
        for i, st in zip(keep, status):      # a refused contig yields no record: say so instead of losing its SVs silently
            if int(st) < 0:
                logging.getLogger("focalsv_amd").error("%s: the aligner refused this contig (status %d); no alignment record, its SVs are not called", names[i], int(st))
            elif int(st) > 0:
                logging.getLogger("focalsv_amd").warning("%s: no chain against its reference window (unaligned)", names[i])
        # read signatures (extract_reads_signature.py): from the records handed in, or straight from the BAM
        if read_records is None and read_bam_file:
            from .. import bam
            rsigs = bam.reads_signatures(ctx, read_bam_file, chrom, 50)
        else:
            rsigs = reads_signature.reads_signatures(read_records or [], 50)
    finally:
        if own:
            ctx.close()
    knames = [names[i] for i in keep]
    records = records_from_alignment(rec, cigar, knames, [chrom] * len(keep), [ref_start[win_index[wins[i]]] for i in keep])
    sig_dir = os.path.join(output_dir, 'signature')
    os.makedirs(sig_dir, exist_ok=True)
    with open(os.path.join(sig_dir, '%s_cigar.txt' % chrom), 'w') as fw:  # CCS.py:726-735
        for r in records:
            s, e = S.get_read_start_end(r.cigar)
            print(r.qname + '\t' + str(r.mapq) + '\t' + str(r.pos) + '\t' + str(r.reference_end) + '\t' + "%d\t%d\t%s\t" % (s, e, '-' if r.is_reverse else '+') + str([tuple(c) for c in r.cigar]), file=fw)
    contigs = dict(zip(names, seqs))
    paired, body = call_chromosome(records, chrom, chrom_seq, contigs, data_type)
    chr_vcf = os.path.join(output_dir, "dippav_variant_chr%d.vcf" % chr_num)
    with open(chr_vcf, 'w') as f:
        f.writelines(header); f.writelines(body)
    raw = os.path.join(output_dir, "dippav_raw_variant.vcf")
    with open(raw, 'w') as f:
        f.writelines(header); f.writelines(body)
    # read signatures -> FP filter -> redundancy
    reads_signature.write_reads_sig(rsigs, output_dir, chrom)
    filtered = os.path.join(output_dir, "dippav_variant_filtered.vcf")
    fp_filter.FP_filter(raw, os.path.join(output_dir, 'reads_signature'), filtered)
    redundancy.remove_redundancy(filtered, os.path.join(output_dir, 'final_vcf'))
    return os.path.join(output_dir, 'final_vcf', 'dippav_variant_no_redundancy.vcf')
