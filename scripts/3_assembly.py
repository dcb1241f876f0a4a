#!/usr/bin/env python3
"""GPU drop-in for focalsv/3_assembly.py: same flags (3_assembly.py:11-17), same region-directory contract."""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focalsv_amd.assembly import assembly, setup_logging  # noqa: E402

parser = ArgumentParser(description="Assemble sequences and call SVs:")
parser.add_argument('--bam_file', '-bam', help="BAM file", required=True)
parser.add_argument('--chr_num', '-chr', type=int, help="Chromosome number for target variant or region", required=True)
parser.add_argument('--ref_file', '-r', help="Reference FASTA file", required=True)
parser.add_argument('--out_dir', '-o', required=True, help="Output directory")
parser.add_argument('--num_threads', '-t_chr', type=int, help="Number of threads, default = 8 (accepted for compatibility)", default=8)
parser.add_argument('--num_cpus', '-t', type=int, help="Number of CPUs, default = 10 (accepted for compatibility)", default=10)
parser.add_argument('--data_type', '-d', type=int, help="HIFI = 0 or CLR = 1 data", default=0)
parser.add_argument('--device', type=int, default=0, help="GPU index (extension)")

if __name__ == "__main__":
    args = parser.parse_args()
    logger = setup_logging("3_ASSEMBLY", args.out_dir)
    logger.info("Starting assembly process (MI355X)")
    if args.data_type != 0:
        logger.warning("CLR/ONT read sets go through the same GPU assembler (the reference uses Flye/Shasta there)")
    st = assembly(args.out_dir, args.num_cpus, args.num_threads, args.data_type, logger, device=args.device)
    bad = {k: v for k, v in st.items() if v}
    if bad:
        logger.warning(f"read sets with a non-zero status: {bad}")
