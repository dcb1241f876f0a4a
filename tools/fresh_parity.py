#!/usr/bin/env python3
"""Exploration aid (not a test): oracle/asm.c against the reference's hifiasm-0.14 (oracle/_ref) on read sets of seeds outside every
golden file -- corrected reads after one, two and three rounds (hifiasm -r N --write-ec) and the contigs.
    python tools/fresh_parity.py <first seed> <count> [depth ...]"""
import hashlib, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from focalsv_amd import synth
from tests import oracle_lib as O
HIFIASM = os.path.join(ROOT, "oracle", "_ref", "hifiasm-0.14")


def canon(s):
    return min(s, synth.revcomp(s))


def hifiasm(reads, rounds, tmp):
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
    ctg = sorted(canon(l.split("\t")[2].strip().encode()) for l in open(os.path.join(tmp, "x.asm.p_ctg.gfa")) if l.startswith("S"))
    return [canon(ec[f"r{j}"]) for j in range(len(reads))], ctg


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    depths = [float(x) for x in sys.argv[3:]] or [15.0]
    bad = n = 0
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(first, first + count):
            depth = depths[i % len(depths)]
            width = tuple(int(x) for x in os.environ.get("FRESH_WIDTHS", "30000,50000,70000").split(","))[i % 3]
            r = synth.make_region(i, width=width, depth_per_hap=depth)
            for h in (1, 2):
                reads = r.reads[h - 1]
                row = []
                for rounds in (1, 2, 3):
                    ref_reads, ref_ctg = hifiasm(reads, rounds, tmp)
                    p = O.default_params()
                    p.n_rounds = rounds
                    ctg, corr = O.assemble(reads, p)
                    nd = sum(canon(c) != e for c, e in zip(corr, ref_reads))
                    row.append(nd)
                    if rounds == 3:
                        row.append("ctg ok" if sorted(canon(c) for c in ctg) == ref_ctg else "CTG DIFF")
                n += 1
                if any(x not in (0, "ctg ok") for x in row):
                    bad += 1
                print(i, h, width, depth, len(reads), "reads differing after 1 / 2 / 3 rounds:", row, flush=True)
    print("sets with a difference:", bad, "of", n)


main()
