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


def main():
    regions = list(range(0, 16)) + [22, 38, 39]
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for i in regions:
            r = synth.make_region(i)
            d = synth.write_region_dir(r, os.path.join(tmp, f"r{i}"))  # fresh dir: hifiasm reloads stale *.bin caches
            for h in (1, 2):
                subprocess.run([HIFIASM, "-f0", "-o", f"PS1_hp{h}.asm", "-t", "8", f"PS1_hp{h}.fa"], cwd=d, check=True,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                seqs = [l.split("\t")[2].strip().encode() for l in open(os.path.join(d, f"PS1_hp{h}.asm.p_ctg.gfa")) if l.startswith("S")]
                hap = r.haps[h - 1]
                out.append({"region": i, "hap": h, "n_reads": len(r.reads[h - 1]),
                            "reads_md5": hashlib.md5(b"\n".join(r.reads[h - 1])).hexdigest(),
                            "hap_len": len(hap), "contig_equals_haplotype": [canon(s) == canon(hap) for s in seqs],
                            "contigs": [{"len": len(s), "md5": hashlib.md5(canon(s)).hexdigest()} for s in seqs]})
                print(i, h, [(len(s), canon(s) == canon(hap)) for s in seqs])
    json.dump({"source": "hifiasm-0.14 -f0 -t 8 via oracle/_ref (reference sources compiled in place)", "sets": out},
              open(os.path.join(ROOT, "tests", "golden", "hifiasm_contigs.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
