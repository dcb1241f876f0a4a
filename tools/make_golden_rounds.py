#!/usr/bin/env python3
"""Per-round golden data: the corrected reads of the reference's hifiasm-0.14 (oracle/_ref) after ONE and after TWO correction
rounds (`-r 1`, `-r 2`; three rounds are in hifiasm_repeats.json / hifiasm_contigs.json already) for the 36 repeat-rich read sets
and a spread of the random-sequence sets.  They pin the restatement round by round: with the second pass over the window
junctions (process_boundary, Correct.cpp:4453) and the haplotype partition the oracle's reads are identical to hifiasm's after
every round, not only at the end -> tests/golden/hifiasm_rounds.json.  Needs /root/reference (oracle/ref.mk)."""
import hashlib, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from focalsv_amd import synth
HIFIASM = os.path.join(ROOT, "oracle", "_ref", "hifiasm-0.14")


def canon(s):
    return min(s, synth.revcomp(s))


def ec_digest(reads, rounds):
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "x.fa"), "w") as f:
            for j, rd in enumerate(reads):
                f.write(f">r{j}\n{rd.decode()}\n")
        subprocess.run([HIFIASM, "-f0", "--write-ec", "-r", str(rounds), "-o", "x.asm", "-t", "8", "x.fa"], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        ec, name = {}, None
        for l in open(os.path.join(tmp, "x.asm.ec.fa")):
            if l.startswith(">"):
                name = l[1:].strip()
            else:
                ec[name] = l.strip().encode()
    return hashlib.md5(b"\n".join(canon(ec[f"r{j}"]) for j in range(len(reads)))).hexdigest()


def main():
    out = []
    for i in range(36):
        reads = synth.make_repeat_region(i).reads[0]
        out.append({"kind": "repeat", "index": i, "reads_md5": hashlib.md5(b"\n".join(reads)).hexdigest(),
                    "round_md5": [ec_digest(reads, 1), ec_digest(reads, 2)]})
        print("repeat", i, flush=True)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "hifiasm_contigs.json")))["sets"]
    for g in gold:
        if g["region"] < 500 and g["region"] not in (0, 1):
            continue
        reads = synth.make_region(g["region"], width=g["width"], depth_per_hap=g["depth"]).reads[g["hap"] - 1]
        assert hashlib.md5(b"\n".join(reads)).hexdigest() == g["reads_md5"]
        out.append({"kind": "region", "region": g["region"], "hap": g["hap"], "width": g["width"], "depth": g["depth"], "reads_md5": g["reads_md5"],
                    "round_md5": [ec_digest(reads, 1), ec_digest(reads, 2)]})
        print("region", g["region"], g["hap"], flush=True)
    json.dump({"source": "tools/make_golden_rounds.py: hifiasm-0.14 (the reference's, built in place) -f0 --write-ec -r 1 / -r 2; md5 of the "
                         "corrected reads (canonical strand) joined by newlines", "sets": out},
              open(os.path.join(ROOT, "tests", "golden", "hifiasm_rounds.json"), "w"), indent=0)


main()
