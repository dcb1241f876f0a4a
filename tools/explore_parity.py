#!/usr/bin/env python3
"""Exploration aid (not a test): oracle/asm.c vs the reference's hifiasm-0.14 (oracle/_ref) on read sets outside the golden
grid -- other widths, depths, error rates.  Prints one line per set; mismatches are candidates for new goldens / fixes."""
import hashlib, os, subprocess, sys, tempfile, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from focalsv_amd import synth
from tests import oracle_lib as O
HIFIASM = os.path.join(ROOT, "oracle", "_ref", "hifiasm-0.14")

def canon(s):
    return min(s, synth.revcomp(s))

def main():
    import numpy as np
    grid = list(itertools.product([14000, 26000, 50000, 100000], [8.0, 15.0, 25.0]))
    n_ok = n = 0
    with tempfile.TemporaryDirectory() as tmp:
        for gi, (width, depth) in enumerate(grid):
            for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
                i = 500 + gi * 10 + rep
                r = synth.make_region(i, width=width, depth_per_hap=depth)
                d = synth.write_region_dir(r, os.path.join(tmp, f"r{i}"))
                for h in (1, 2):
                    subprocess.run([HIFIASM, "-f0", "-o", f"PS1_hp{h}.asm", "-t", "8", f"PS1_hp{h}.fa"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                    ref = sorted(canon(l.split("\t")[2].strip().encode()) for l in open(os.path.join(d, f"PS1_hp{h}.asm.p_ctg.gfa")) if l.startswith("S"))
                    mine, _ = O.assemble(r.reads[h - 1], O.default_params())
                    mine = sorted(canon(c) for c in mine)
                    ok = mine == ref
                    n += 1; n_ok += ok
                    print(i, width, depth, h, len(r.reads[h - 1]), "OK" if ok else "DIFF", [len(c) for c in ref], [len(c) for c in mine], flush=True)
    print("match", n_ok, "of", n)

main()
