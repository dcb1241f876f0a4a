#!/usr/bin/env python3
"""Mint contig-level golden data from the reference assembler itself.

For seeded synthetic regions (focalsv_amd/synth.py) run the reference's hifiasm-0.14
(built by oracle/ref.mk from /root/reference/software/hifiasm-0.14, invoked as
focalsv/3_assembly/run_assembly.py:21 does, plus -f0 which only skips the 16 GiB
Bloom filter -- contigs are byte-identical) on each PS1_hp{1,2}.fa and
record the length and md5 of every primary contig in canonical orientation
(min(seq, revcomp)).  Only these digests are committed (tests/golden/hifiasm_contigs.json).
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from focalsv_amd import synth  # noqa: E402

HIFIASM = os.path.join(ROOT, "oracle", "_ref", "hifiasm-0.14")


def canon(seq: bytes) -> bytes:
    rc = synth.revcomp(seq)
    return min(seq, rc)


def run_set(tmp, tag, r, h):
    """hifiasm-0.14 on one read set: primary contigs and (--write-ec) the corrected reads"""
    d = synth.write_region_dir(r, os.path.join(tmp, tag))  # fresh dir: hifiasm reloads stale *.bin caches
    subprocess.run([HIFIASM, "-f0", "--write-ec", "-o", f"PS1_hp{h}.asm", "-t", "8", f"PS1_hp{h}.fa"], cwd=d, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    seqs = [l.split("\t")[2].strip().encode() for l in open(os.path.join(d, f"PS1_hp{h}.asm.p_ctg.gfa")) if l.startswith("S")]
    ec, name = {}, None
    for l in open(os.path.join(d, f"PS1_hp{h}.asm.ec.fa")):
        if l.startswith(">"):
            name = l[1:].strip(); ec[name] = []
        else:
            ec[name].append(l.strip())
    reads = r.reads[h - 1]
    corrected = [canon("".join(ec[f"r{r.index}_h{h}_{j}"]).encode()) for j in range(len(reads))]
    return seqs, hashlib.md5(b"\n".join(corrected)).hexdigest()


def main():
    # (region index, window width, depth per haplotype): the bench geometry, then other widths / depths (8x: some reads keep
    # errors and the layout needs inexact overlaps; 14 kb: chains of fewer than four reads, hifiasm writes no contig)
    grid = [(i, 50000, 15.0) for i in list(range(0, 16)) + [22, 38, 39]]
    for gi, (width, depth) in enumerate((w, d) for w in (14000, 26000, 50000, 100000) for d in (8.0, 15.0, 25.0)):
        grid += [(500 + gi * 10 + rep, width, depth) for rep in range(2)]
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, width, depth in grid:
            r = synth.make_region(i, width=width, depth_per_hap=depth)
            for h in (1, 2):
                seqs, ec_md5 = run_set(tmp, f"r{i}_{h}", r, h)
                hap = r.haps[h - 1]
                out.append({"region": i, "hap": h, "width": width, "depth": depth, "n_reads": len(r.reads[h - 1]),
                            "reads_md5": hashlib.md5(b"\n".join(r.reads[h - 1])).hexdigest(),
                            "hap_len": len(hap), "contig_equals_haplotype": [canon(s) == canon(hap) for s in seqs],
                            "contigs": [{"len": len(s), "md5": hashlib.md5(canon(s)).hexdigest()} for s in seqs],
                            "corrected_reads_md5": ec_md5})
                print(i, width, depth, h, [(len(s), canon(s) == canon(hap)) for s in seqs], flush=True)
    json.dump({"source": "hifiasm-0.14 -f0 --write-ec -t 8 via oracle/_ref (reference sources compiled in place); corrected_reads_md5 = md5 of the "
                         "corrected reads (each as min(seq, revcomp)) joined by newlines, in input order",
               "sets": out}, open(os.path.join(ROOT, "tests", "golden", "hifiasm_contigs.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
