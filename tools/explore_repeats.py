#!/usr/bin/env python3
"""Exploration aid (not a test): oracle/asm.c vs the reference's hifiasm-0.14 (oracle/_ref, --write-ec) on read sets whose
haplotype carries interspersed repeats -- copies of one 0.3 - 6 kb element at several places, 0 - 5 % diverged from each other
(synth.make_repeat_region).  Prints per set whether contigs and corrected reads are identical."""
import hashlib, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from focalsv_amd import synth
from tests import oracle_lib as O
HIFIASM = os.path.join(ROOT, "oracle", "_ref", "hifiasm-0.14")


def canon(s):
    return min(s, synth.revcomp(s))


def ec_reads(path):
    out, name = {}, None
    for l in open(path):
        if l.startswith(">"):
            name = l[1:].strip()
        else:
            out[name] = l.strip().encode()
    return out


def main():
    n0, n1 = int(sys.argv[1]), int(sys.argv[2])
    ok_c = ok_r = n = 0
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(n0, n1):
            r = synth.make_repeat_region(i)
            d = os.path.join(tmp, f"r{i}")
            os.makedirs(d)
            with open(os.path.join(d, "x.fa"), "w") as f:
                for j, rd in enumerate(r.reads[0]):
                    f.write(f">r{j}\n{rd.decode()}\n")
            p = subprocess.run([HIFIASM, "-f0", "--write-ec", "-o", "x.asm", "-t", "8", "x.fa"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
            info = [l for l in p.stderr.splitlines() if "peak_hom" in l or "filtered out" in l]
            ref = sorted(canon(l.split("\t")[2].strip().encode()) for l in open(os.path.join(d, "x.asm.p_ctg.gfa")) if l.startswith("S"))
            ec = ec_reads(os.path.join(d, "x.asm.ec.fa"))
            mine, corr = O.assemble(r.reads[0], O.default_params())
            mine = sorted(canon(c) for c in mine)
            same_reads = sum(1 for j, c in enumerate(corr) if ec.get(f"r{j}") == c)
            c_ok = mine == ref
            n += 1; ok_c += c_ok; ok_r += same_reads == len(corr)
            print(i, r.note, "reads", len(corr), "contigs", "OK" if c_ok else "DIFF %s vs %s" % ([len(c) for c in ref], [len(c) for c in mine]),
                  "| corrected reads identical %d/%d" % (same_reads, len(corr)), "|", "; ".join(x.split("] ", 1)[-1] for x in info[-2:]), flush=True)
    print("contigs", ok_c, "of", n, "; corrected read sets", ok_r, "of", n)


main()
