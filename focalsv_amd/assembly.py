"""Step 3 of FocalSV on the GPU: mirrors focalsv/3_assembly.py + 3_assembly/{run_assembly,post_assembly,combine_fas}.py.

Reference contract kept: for every `<out_dir>/regions/Region*/` directory each read FASTA `X.fa` becomes an assembly
`X.asm.p_ctg.gfa.fa` (HiFi naming), and `HP1.fa` / `HP2.fa` are the concatenations of the `*hp1*` / `*hp2*` assemblies
(combine_fas.py:10-35).  Instead of one hifiasm process per FASTA (run_assembly.py:15-44) all read sets of all
regions go through one fsv_assemble_batch call.  Failures never abort the batch: a set that cannot be assembled
yields an empty HP file and a log line, as the reference's ignored exit codes do (run_assembly.py:25).
"""
import logging
import os
from typing import Dict, List, Optional

from . import _lib, fasta
from .readsets import pack_sets


def setup_logging(step_name, out_dir):
    """same log layout as focalsv/utils.py:6-20"""
    log_dir = os.path.join(out_dir, "log")
    os.makedirs(log_dir, exist_ok=True)
    logger = logging.getLogger(step_name)
    if not logger.handlers:
        h = logging.FileHandler(os.path.join(log_dir, f"{step_name}.log"), mode='a')
        h.setFormatter(logging.Formatter('%(asctime)s - %(levelname)s - %(message)s'))
        logger.addHandler(h)
        logger.setLevel(logging.INFO)
    return logger


def find_read_sets(regions_dir: str, data_type: int = 0) -> List[str]:
    """HiFi: every *.fa of every Region* dir (run_assembly.py:33-38); CLR/ONT: only PS*.fa (run_assembly.py:52-54)."""
    out = []
    for fd in sorted(os.listdir(regions_dir)):
        if not fd.startswith("Region"):
            continue
        d = os.path.join(regions_dir, fd)
        for f in sorted(os.listdir(d)):
            if not f.endswith(".fa") or f in ("HP1.fa", "HP2.fa") or f.endswith(".gfa.fa"):
                continue
            if data_type != 0 and not f.startswith("PS"):
                continue
            out.append(os.path.join(d, f))
    return out


def assembly(out_dir: str, cpu: int = 10, threads: int = 8, data_type: int = 0, logger=None, ctx: Optional[_lib.Context] = None,
             device: int = 0, skip_existing: bool = True) -> Dict[str, int]:
    """3_assembly.py:28-41.  cpu/threads are accepted for CLI compatibility (the GPU batch replaces both)."""
    logger = logger or setup_logging("3_ASSEMBLY", out_dir)
    regions_dir = os.path.join(out_dir, "regions")
    fas = find_read_sets(regions_dir, data_type)
    if skip_existing:  # checkpoint/resume at the reference's granularity: regions with HP files are done
        fas = [f for f in fas if not (os.path.exists(os.path.join(os.path.dirname(f), "HP1.fa")) and os.path.exists(os.path.join(os.path.dirname(f), "HP2.fa")))]
    logger.info(f"read sets to assemble: {len(fas)}")
    status: Dict[str, int] = {}
    if fas:
        if any('unphased' in os.path.basename(f) for f in fas):
            logger.warning("unphased read sets are assembled as a single haplotype (dual-haplotype partition is not implemented)")
        sets = [fasta.read_reads(f) for f in fas]
        own = ctx is None
        ctx = ctx or _lib.Context(device)
        try:
            b = pack_sets(sets)
            d = ctx.upload(b.words)
            try:
                contigs, cset, cnr, st = ctx.assemble_batch(d, b.word_off, b.read_len, b.set_start)
            finally:
                ctx.dev_free(d)
        finally:
            if own:
                ctx.close()
        for si, f in enumerate(fas):
            outp = f[:-3] + ".asm.p_ctg.gfa.fa"
            fasta.write_contig_fasta(outp, outp, [c for c, s in zip(contigs, cset) if s == si])
            status[f] = int(st[si])
            if st[si]:
                logger.warning(f"{f}: assembly status {int(st[si])}")
    combine_fas(regions_dir, logger)
    return status


def combine_fas(regions_dir: str, logger=None):
    """combine_fas.py:10-35 (HiFi naming: *hp1.asm.p_ctg.gfa.fa / *hap1.p_ctg.gfa.fa)"""
    for fd in sorted(os.listdir(regions_dir)):
        d = os.path.join(regions_dir, fd)
        if not fd.startswith("Region") or not os.path.isdir(d):
            continue
        for hp, tags in ((1, ("hp1.asm.p_ctg.gfa.fa", "hap1.p_ctg.gfa.fa")), (2, ("hp2.asm.p_ctg.gfa.fa", "hap2.p_ctg.gfa.fa"))):
            parts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(tags)]
            with open(os.path.join(d, f"HP{hp}.fa"), 'w') as out:
                for p in parts:
                    with open(p) as f:
                        out.write(f.read())
            if logger:
                logger.info(f"*** Finished combining HP{hp} for {d} ***")
