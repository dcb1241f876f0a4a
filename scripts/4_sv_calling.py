#!/usr/bin/env python3
"""GPU drop-in for focalsv/4_sv_calling.sh (the live entry) with the flags of focalsv/4_sv_calling.py:11-18.

For one chromosome: gathers regions/*/HP1.fa and HP2.fa (4_sv_calling.sh:17-18) keeping the region tag of every
contig, then runs dippav_variant_call (contig alignment on the GPU, signature extraction, FP filter, redundancy
removal).  Output: <out_dir>/SV/chr<N>/final_vcf/dippav_variant_no_redundancy.vcf like the reference."""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focalsv_amd import fasta  # noqa: E402
from focalsv_amd.assembly import setup_logging  # noqa: E402
from focalsv_amd.dippav.variant_call import dippav_variant_call  # noqa: E402

parser = ArgumentParser(description="Assemble sequences:")
parser.add_argument('--bam_file', '-bam', help="BAM file", required=True)
parser.add_argument('--chr_num', '-chr', type=int, help="Chromosome number for target variant or region", required=True)
parser.add_argument('--reference', '-r', help="Reference fasta file", required=True)
parser.add_argument('--out_dir', '-o', help="Directory to store assembly results", default="./RegionBased_results")
parser.add_argument('--num_threads', '-t_chr', type=int, default=8)
parser.add_argument('--num_cpus', '-t', type=int, default=10)
parser.add_argument('--log_dir', '-log', help="Position of log directory", default=None)
parser.add_argument('--data_type', '-d', type=int, help="HIFI = 0 or CLR = 1 or ONT = 2 data", default=0)
parser.add_argument('--device', type=int, default=0, help="GPU index (extension)")


def gather_haplotype_fasta(regions_dir, hp, out_path):
    """`cat regions/*/HP<hp>.fa` with every header replaced by '<Region tag> <n>' so that the window survives"""
    n = 0
    with open(out_path, 'w') as out:
        for fd in sorted(os.listdir(regions_dir)):
            p = os.path.join(regions_dir, fd, "HP%d.fa" % hp)
            if not fd.startswith("Region") or not os.path.exists(p):
                continue
            for _, seq in fasta.read_fasta(p):
                n += 1
                out.write(">%s_a_hp%d_%d\n%s\n" % (fd, hp, n, fasta.fold(seq)))
    return n


if __name__ == "__main__":
    args = parser.parse_args()
    logger = setup_logging("4_SV_CALLING", args.out_dir)
    chrom_dir = os.path.join(args.out_dir, "SV", "chr%d" % args.chr_num)
    os.makedirs(chrom_dir, exist_ok=True)
    hp1 = os.path.join(args.out_dir, "chr%d_HP1_new.fa" % args.chr_num)
    hp2 = os.path.join(args.out_dir, "chr%d_HP2_new.fa" % args.chr_num)
    n1 = gather_haplotype_fasta(os.path.join(args.out_dir, "regions"), 1, hp1)
    n2 = gather_haplotype_fasta(os.path.join(args.out_dir, "regions"), 2, hp2)
    logger.info(f"contigs: hp1 {n1}, hp2 {n2}")
    dtype = {0: "CCS", 1: "CLR", 2: "ONT"}[args.data_type]
    final = dippav_variant_call(dtype, args.bam_file, args.reference, hp1, hp2, chrom_dir, args.chr_num, None, args.num_cpus, device=args.device)
    logger.info(f"final VCF: {final}")
    print(final)
