#!/usr/bin/env python3
"""Read-level golden data for UNPHASED read sets: both haplotypes' reads of regions 100 .. 147 (the seeds of the fresh block in
hifiasm016_unphased.json) in one set through the reference's hifiasm-0.14 (oracle/_ref), corrected reads after one, two and three
rounds (-r N --write-ec) -> tests/golden/hifiasm_mixed_reads.json.  In such a set the haplotype partition (partition_overlaps_advance)
decides at every heterozygous site which overlaps may vote: this pins it read for read, where hifiasm016_unphased.json pins the contigs.
Needs /root/reference (oracle/ref.mk)."""
import hashlib, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from focalsv_amd import synth  # noqa: E402
HIFIASM = os.path.join(ROOT, "oracle", "_ref", "hifiasm-0.14")


def canon(s):
    return min(s, synth.revcomp(s))


def corrected_md5(reads, rounds, tmp):
    for f in os.listdir(tmp):
        os.unlink(os.path.join(tmp, f))
    with open(os.path.join(tmp, "x.fa"), "w") as f:
        for j, rd in enumerate(reads):
            f.write(f">r{j}\n{rd.decode()}\n")
    subprocess.run([HIFIASM, "-f0", "--write-ec", "-r", str(rounds), "-o", "x.asm", "-t", "8", "x.fa"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ec, name = {}, None
    for l in open(os.path.join(tmp, "x.asm.ec.fa")):
        if l.startswith(">"):
            name = l[1:].strip()
        else:
            ec[name] = l.strip().encode()
    return hashlib.md5(b"\n".join(canon(ec[f"r{j}"]) for j in range(len(reads)))).hexdigest()


def main():
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(100, 148):
            r = synth.make_region(i)
            reads = r.reads[0] + r.reads[1]
            out.append({"region": i, "n_reads": len(reads), "reads_md5": hashlib.md5(b"\n".join(reads)).hexdigest(),
                        "round_md5": [corrected_md5(reads, rounds, tmp) for rounds in (1, 2, 3)]})
            print(i, len(reads), out[-1]["round_md5"], flush=True)
    json.dump({"source": "tools/make_golden_mixed.py: hifiasm-0.14 (the reference's, built in place) -f0 --write-ec -r 1 / 2 / 3 on both haplotypes' reads "
                         "in one set (hp1 first); md5 of the corrected reads (canonical strand) joined by newlines", "sets": out},
              open(os.path.join(ROOT, "tests", "golden", "hifiasm_mixed_reads.json"), "w"), indent=0)


main()
