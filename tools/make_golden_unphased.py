#!/usr/bin/env python3
"""Unphased (diploid) read sets through the reference's hifiasm-0.16.1 (oracle/_ref, what run_assembly.py:17-21 runs on
unphased.fa): both haplotypes' reads of a seeded region in one FASTA -> digests of the bp.hap1 / bp.hap2 primary contigs
-> tests/golden/hifiasm016_unphased.json.  Also the homozygous case (one haplotype's reads only)."""
import hashlib, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from focalsv_amd import synth  # noqa: E402
H16 = os.path.join(ROOT, "oracle", "_ref", "hifiasm-0.16.1")


def canon(s):
    return min(s, synth.revcomp(s))


def main():
    out = []
    # the first twelve are the sets of round 2; regions 100 .. 147 are a run of seeds taken as they come (round 3: whatever they show
    # -- tests/test_oracle_asm.py lists the ones that differ)
    for region, mode in [(i, "mixed") for i in (0, 1, 2, 3, 5, 7, 8, 12, 22, 38)] + [(0, "hp1"), (7, "hp2")] + [(i, "mixed") for i in range(100, 148)]:
        r = synth.make_region(region)
        reads = r.reads[0] + r.reads[1] if mode == "mixed" else r.reads[0 if mode == "hp1" else 1]
        with tempfile.TemporaryDirectory() as tmp:
            with open(os.path.join(tmp, "unphased.fa"), "w") as f:
                for j, rd in enumerate(reads):
                    f.write(">u%d\n%s\n" % (j, rd.decode()))
            subprocess.run([H16, "-f0", "-o", "unphased.asm", "-t", "8", "unphased.fa"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            haps = {}
            for h in (1, 2):
                seqs = [l.split("\t")[2].strip().encode() for l in open(os.path.join(tmp, "unphased.asm.bp.hap%d.p_ctg.gfa" % h)) if l.startswith("S")]
                haps["hap%d" % h] = [{"len": len(s), "md5": hashlib.md5(canon(s)).hexdigest()} for s in seqs]
        out.append({"region": region, "mode": mode, "n_reads": len(reads), "reads_md5": hashlib.md5(b"\n".join(reads)).hexdigest(), **haps})
        print(region, mode, haps, flush=True)
    json.dump({"source": "hifiasm-0.16.1 -f0 -t 8 via oracle/_ref on unphased.fa (both haplotypes' reads, hp1 first)", "sets": out},
              open(os.path.join(ROOT, "tests", "golden", "hifiasm016_unphased.json"), "w"), indent=0)


main()
