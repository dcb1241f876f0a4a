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



def _describe_status(st: int) -> str:
    """the warning bits of a read set's status in words (bit 1 / 2: a read longer than the minimizer / anchor lists hold lost its
    tail -- with the ONT / CLR parameter set that is a read above ~32 kb: include/focalsv_hip.h, fsv_asm_ont_params)"""
    from .pipeline import describe_set_status
    return describe_set_status(st)

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


def _set_cost(reads) -> int:
    """device bytes a read set needs, dominated by the window-task bound: every read against every other, 176 B per window"""
    n = len(reads)
    win = sum((len(r) + 374) // 375 for r in reads)
    return (n - 1) * win * 176 + n * n * 72 + sum(len(r) for r in reads) * 24 if n > 1 else 4096


def assemble_sets(ctx: _lib.Context, sets, logger=None, budget_bytes: Optional[int] = None, set_flags=None, params=None):
    """all read sets of a chromosome through fsv_assemble_batch, in as few batches as the device memory allows
    (FSV_BATCH_GB overrides the default budget of 64 GB of workspace per batch); a batch the library refuses as too large is
    halved and retried.  -> [(contigs, status)] per set, in input order."""
    if budget_bytes is None:
        budget_bytes = int(float(os.environ.get("FSV_BATCH_GB", "64")) * (1 << 30))
    out = [None] * len(sets)

    def run(idx):
        b = pack_sets([sets[i] for i in idx])
        d = ctx.upload(b.words)
        try:
            contigs, cset, cnr, st = ctx.assemble_batch(d, b.word_off, b.read_len, b.set_start, params,
                                                         None if set_flags is None else [set_flags[i] for i in idx])
        finally:
            ctx.dev_free(d)
        per = [[] for _ in idx]
        for c, s in zip(contigs, cset):
            per[int(s)].append(c)
        for k, i in enumerate(idx):
            out[i] = (per[k], int(st[k]))

    def run_or_split(idx):
        try:
            run(idx)
        except _lib.FsvError as e:
            if e.code not in (-3, -5, -6) or len(idx) == 1:   # FSV_ENOMEM, FSV_ECAP, FSV_EUNSUP: too much for one batch
                raise
            if logger:
                logger.warning(f"batch of {len(idx)} read sets refused ({e}); halving")
            run_or_split(idx[: len(idx) // 2])
            run_or_split(idx[len(idx) // 2:])

    batch, used = [], 0
    for i, rs in enumerate(sets):
        c = _set_cost(rs)
        if batch and used + c > budget_bytes:
            run_or_split(batch)
            batch, used = [], 0
        batch.append(i)
        used += c
    if batch:
        run_or_split(batch)
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
    if fas and data_type != 0:
        # CLR / ONT: the reference runs Flye on every PS*.fa (run_assembly.py:46-100, output <X>_flye/assembly.fasta; ONT falls back to
        # Shasta, post_assembly.py:43-76).  Here the same assembler as for HiFi runs with the error model opened up
        # (fsv_asm_clr_params / fsv_asm_ont_params: wide-band K5 / K6) and leaves its contigs where combine_fas_clr / _ont look for Flye's.
        # Checkpoint as the reference's: a set whose assembly.fasta exists is skipped (run_assembly.py:65).
        fas = [f for f in fas if not os.path.exists(os.path.join(f[:-3] + "_flye", "assembly.fasta"))] if skip_existing else fas
        sets = [fasta.read_reads(f) for f in fas]
        own = ctx is None
        ctx = ctx or _lib.Context(device)
        try:
            per_set = assemble_sets(ctx, sets, logger, params=ctx.clr_asm_params() if data_type == 1 else ctx.ont_asm_params())
        finally:
            if own:
                ctx.close()
        # the noisy-read profiles hold 4 096 minimizers per read (k = w = 15: about 32 kb) and 4 096 anchors per pair; a longer read keeps
        # its windows but loses the anchors beyond that (set status bits 1 / 2): say how many reads of a set that concerns
        LONG = 32000
        for f, rs in zip(fas, sets):
            n_long = sum(1 for r_ in rs if len(r_) > LONG)
            if n_long:
                logger.warning(f"{f}: {n_long} of {len(rs)} reads are longer than {LONG} bases: their overlaps are seeded from their first 4 096 minimizers only")
        for f, (contigs, st) in zip(fas, per_set):
            d = f[:-3] + "_flye"
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "assembly.fasta"), "w") as fw:       # Flye's layout: >contig_<n>, sequence folded at 60 columns
                for n, c in enumerate(contigs, 1):
                    t = c.decode()
                    fw.write(">contig_%d\n" % n + "\n".join(t[i:i + 60] for i in range(0, len(t), 60)) + "\n")
            status[f] = int(st)
            if st:
                logger.warning(f"{f}: assembly status {int(st)} ({_describe_status(int(st))})")
        combine_fas(regions_dir, logger, data_type)
        return status
    if fas:
        if any('unphased' in os.path.basename(f) for f in fas):
            logger.info("unphased read sets go through the haplotype partition (FSV_SET_UNPHASED)")
        sets = [fasta.read_reads(f) for f in fas]
        own = ctx is None
        ctx = ctx or _lib.Context(device)
        try:
            per_set = assemble_sets(ctx, sets, logger, set_flags=[_lib.SET_UNPHASED if 'unphased' in os.path.basename(f) else 0 for f in fas])
        finally:
            if own:
                ctx.close()
        for f, (contigs, st) in zip(fas, per_set):
            if 'unphased' in os.path.basename(f):
                # the reference runs hifiasm-0.16.1 here, whose bp.hap1 / bp.hap2 contigs combine_fas puts into HP1 / HP2
                # (run_assembly.py:17-21, combine_fas.py:13-14).  The set was assembled in the unphased mode: two contigs are the
                # two haplotypes; a single contig means no heterozygosity, and 0.16.1 then reports it for both haplotypes
                # (checked against oracle/_ref/hifiasm-0.16.1).  More fragments alternate between the two files.
                haps = ([contigs[0]], [contigs[0]]) if len(contigs) == 1 else (contigs[0::2], contigs[1::2])
                for hap in (1, 2):
                    outp = f[:-3] + ".asm.bp.hap%d.p_ctg.gfa.fa" % hap
                    fasta.write_contig_fasta(outp, outp, haps[hap - 1])
            else:
                outp = f[:-3] + ".asm.p_ctg.gfa.fa"
                fasta.write_contig_fasta(outp, outp, contigs)
            status[f] = int(st)
            if st:
                logger.warning(f"{f}: assembly status {int(st)} ({_describe_status(int(st))})")
    combine_fas(regions_dir, logger)
    return status


def combine_fas(regions_dir: str, logger=None, data_type: int = 0):
    """combine_fas.py:10-35 (HiFi naming: *hp1.asm.p_ctg.gfa.fa / *hap1.p_ctg.gfa.fa); CLR / ONT: the <X>hp1_flye/assembly.fasta of
    combine_fas_clr / combine_fas_ont (:37-111; no Shasta directories are written here)"""
    for fd in sorted(os.listdir(regions_dir)):
        d = os.path.join(regions_dir, fd)
        if not fd.startswith("Region") or not os.path.isdir(d):
            continue
        for hp, tags in ((1, ("hp1.asm.p_ctg.gfa.fa", "hap1.p_ctg.gfa.fa")), (2, ("hp2.asm.p_ctg.gfa.fa", "hap2.p_ctg.gfa.fa"))):
            if data_type != 0:
                parts = [os.path.join(d, f, "assembly.fasta") for f in sorted(os.listdir(d)) if f.endswith("hp%d_flye" % hp)]
                parts = [p for p in parts if os.path.exists(p)]
            else:
                parts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(tags)]
            with open(os.path.join(d, f"HP{hp}.fa"), 'w') as out:
                for p in parts:
                    with open(p) as f:
                        out.write(f.read())
            if logger:
                logger.info(f"*** Finished combining HP{hp} for {d} ***")
