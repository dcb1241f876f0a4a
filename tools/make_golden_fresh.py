#!/usr/bin/env python3
"""Golden data from read sets of seeds no other golden file uses (7000 ..., 8000 ..., 9000 ...: 14 - 100 kb windows, 6x - 40x per haplotype): the corrected
reads of the reference's hifiasm-0.14 (oracle/_ref) after one, two and three correction rounds (-r N --write-ec) and its contigs
-> tests/golden/hifiasm_fresh.json.  Three of these sets showed what the other goldens did not (a 300-base overlap voting at a read's
end; the two directions of a gapped final overlap differing by an indel near a read end; an overlap that only the left-extension
rescue pass accepts).  Needs /root/reference (oracle/ref.mk)."""
import hashlib, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from focalsv_amd import synth
HIFIASM = os.path.join(ROOT, "oracle", "_ref", "hifiasm-0.14")
BLOCKS = ((7000, 20, (15.0, 10.0, 25.0, 8.0), (30000, 50000, 70000)), (8000, 20, (12.0, 20.0, 30.0, 6.0), (30000, 50000, 70000)),
          (9000, 18, (40.0, 9.0, 18.0, 7.0), (14000, 26000, 100000)))      # (first seed, count, depths per haplotype, window widths)


def canon(s):
    return min(s, synth.revcomp(s))


def run(reads, rounds, tmp):
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
    return hashlib.md5(b"\n".join(canon(ec[f"r{j}"]) for j in range(len(reads)))).hexdigest(), ctg


def main():
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, depths, widths in ((i, d, w) for first, count, d, w in BLOCKS for i in range(first, first + count)):
            depth, width = depths[i % len(depths)], widths[i % len(widths)]
            r = synth.make_region(i, width=width, depth_per_hap=depth)
            for h in (1, 2):
                reads = r.reads[h - 1]
                md5s, ctg = [], None
                for rounds in (1, 2, 3):
                    m, ctg = run(reads, rounds, tmp)
                    md5s.append(m)
                out.append({"region": i, "hap": h, "width": width, "depth": depth, "reads_md5": hashlib.md5(b"\n".join(reads)).hexdigest(),
                            "round_md5": md5s, "contigs": [[len(c), hashlib.md5(c).hexdigest()] for c in ctg]})
                print(i, h, width, depth, len(reads), [len(c) for c in ctg], flush=True)
    json.dump({"source": "tools/make_golden_fresh.py: hifiasm-0.14 (the reference's, built in place) -f0 --write-ec -r 1 / 2 / 3; md5 of the corrected "
                         "reads (canonical strand) joined by newlines; contigs of the three-round run (length, md5 of the canonical strand)", "sets": out},
              open(os.path.join(ROOT, "tests", "golden", "hifiasm_fresh.json"), "w"), indent=0)


main()
