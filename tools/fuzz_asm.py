#!/usr/bin/env python3
"""Exploration aid: random read sets of many shapes (window 14-60 kb, 6-30x per haplotype, phased / both haplotypes mixed / uneven
mixes, tandem-repeat regions) through fsv_assemble_batch in one call and through oracle/asm.c set by set; reports every set whose
corrected reads or contigs differ.  Run on a GPU box:  python tools/fuzz_asm.py [n_sets] [seed]"""
import os, random, sys
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from focalsv_amd import synth


def make_sets(n, seed):
    rng = random.Random(seed)
    sets, notes = [], []
    for i in range(n):
        width = rng.choice((14000, 20000, 26000, 40000, 60000))
        depth = rng.choice((6.0, 8.0, 10.0, 15.0, 22.0, 30.0))
        region = rng.randrange(3000, 9000)
        r = synth.make_region(region, width=width, depth_per_hap=depth)
        kind = rng.choice(("hp1", "hp2", "mixed", "uneven"))
        if kind == "hp1":
            s = r.reads[0]
        elif kind == "hp2":
            s = r.reads[1]
        elif kind == "mixed":
            s = r.reads[0] + r.reads[1]
        else:
            s = r.reads[0] + [x for x in r.reads[1] if rng.random() < 0.3]
        sets.append(s)
        notes.append((region, width, depth, kind, len(s)))
    return sets, notes


def oracle_one(s):
    from tests import oracle_lib as O
    return O.assemble(s, O.default_params())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from focalsv_amd import _lib
    from tests.test_gpu_asm import gpu_assemble
    sets, notes = make_sets(n, seed)
    ctx = _lib.Context(0)
    contigs, cset, status, reads, b = gpu_assemble(ctx, sets)
    ctx.close()
    with ProcessPoolExecutor(max_workers=12) as ex:
        ora = list(ex.map(oracle_one, sets))
    k, bad = 0, 0
    for si, s in enumerate(sets):
        oc, ocorr = ora[si]
        dr = [j for j in range(len(s)) if reads[k + j] != ocorr[j]]
        k += len(s)
        mine = [c for c, cs in zip(contigs, cset) if cs == si]
        if dr or mine != oc:
            bad += 1
            print("DIFF", si, notes[si], "reads", dr[:6], "contigs", [len(c) for c in mine], [len(c) for c in oc], "status", int(status[si]), flush=True)
    for si in range(n):
        if int(status[si]) & 64:
            print("W_SITES", si, notes[si], flush=True)
    print("sets", n, "differing", bad, "statuses", sorted(set(int(x) for x in status)))


if __name__ == "__main__":
    main()
