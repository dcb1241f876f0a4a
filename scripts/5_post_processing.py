#!/usr/bin/env python3
"""Drop-in for focalsv/5_post_processing/FocalSV_Filter_GT_Correct.py (same flags): read signatures (extracted from the BAM, or
--sigdir holding DEL.sigs / INS.sigs), signature support per call, empirical support filter, then genotype correction (HiFi) or the
genotypes / insertions of the read-based draft calls (CLR, ONT: clustered and genotyped from the read signatures, -r needed)
-> <out_dir>/FocalSV_Final_SV.vcf.  The read BAM is read with the library's own reader (no pysam / samtools)."""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focalsv_amd import post_processing  # noqa: E402

parser = ArgumentParser(description="Filters SVs from a VCF file by read-signature support and corrects genotypes.")
parser.add_argument("--bam_file", '-bam', type=str, required=True, help="whole genome reads BAM file")
parser.add_argument("--data_type", '-d', type=str, choices=["ONT", "Hifi", "HIFI", "CLR"], default="Hifi")
parser.add_argument("--ref_file", '-r', type=str, help="reference FASTA (CLR / ONT: REF / ALT alleles of the read-based draft calls)")
parser.add_argument("--chr_num", '-chr', type=str, choices=[str(i) for i in range(1, 23)] + ['wgs'], required=True)
parser.add_argument("--out_dir", '-o', type=str, default="./FocalSV_Result")
parser.add_argument("--num_threads", '-thread', type=int, default=10)
parser.add_argument("--sigdir", '-sig', type=str, help="pre-extracted reads signature directory (DEL.sigs, INS.sigs); extracted from the BAM when absent")
parser.add_argument("--draft_vcf", type=str, default=None,
                    help="CLR / ONT: ready-made read-based draft calls (default <sigdir>/reads_draft_variants.vcf, made when absent) (extension)")

if __name__ == "__main__":
    args = parser.parse_args()
    dtype = "Hifi" if args.data_type == "HIFI" else args.data_type
    print(post_processing.filter_gt_correct(args.bam_file, args.out_dir, args.chr_num, args.sigdir, dtype, args.draft_vcf, args.ref_file))
