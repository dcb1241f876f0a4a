"""fsv_align_batch / fsv_nw on the GPU vs the CPU oracle (bit-exact CIGARs) and vs the in-tree ksw2 golden."""
import json
import os
import random
import re

import numpy as np
import pytest

from focalsv_amd import _lib, synth
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    with _lib.Context(0) as c:
        yield c


def test_nw_matches_ksw2_golden(ctx, golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "ksw_extz2.json")))["cases"]
    checked = 0
    for c in cases:
        ops = re.findall(r"(\d+)([MID])", c["cigar"])
        if sum(int(n) for n, o in ops if o in "MI") != len(c["query"]) or sum(int(n) for n, o in ops if o in "MD") != len(c["target"]):
            continue
        p = ctx.default_aln_params()
        p.a, p.b, p.q, p.e, p.q2, p.e2 = c["a"], c["b"], c["q"], c["e"], -1, -1
        sc, cg = ctx.nw(c["target"].encode(), c["query"].encode(), p)
        assert sc == c["score"] and O.cigar_str(cg) == c["cigar"], c
        checked += 1
    assert checked >= 40


def test_nw_random_dual_affine_vs_oracle(ctx):
    rng = random.Random(3)
    for it in range(60):
        tl = rng.choice([1, 2, 17, 80, 300, 1500, 2600])
        t = "".join(rng.choice("ACGT") for _ in range(tl))
        q = list(t)
        for _ in range(rng.randint(0, 6)):
            if not q:
                break
            a = rng.randrange(len(q))
            k = rng.choice([1, 1, 3, 40, 200])
            if rng.random() < 0.5:
                del q[a:a + k]
            else:
                q[a:a] = [rng.choice("ACGT") for _ in range(k)]
        q = "".join(q) or "A"
        sc, cg = ctx.nw(t.encode(), q.encode())
        osc, ocg = O.nw(t.encode(), q.encode())
        assert sc == osc and list(cg) == list(ocg), (it, tl, len(q), O.cigar_str(cg), O.cigar_str(ocg))


def test_align_batch_matches_oracle(ctx):
    regions = [synth.make_region(i) for i in (0, 4, 7, 15)]
    contigs, cref, refs = [], [], []
    for ri, r in enumerate(regions):
        refs.append(r.ref)
        for h in (0, 1):
            contigs.append(r.haps[h]); cref.append(ri)
            contigs.append(synth.revcomp(r.haps[h])); cref.append(ri)
    rec, cigar, status = ctx.align_batch(contigs, cref, refs)
    assert (status == 0).all() and len(rec) == len(contigs)
    for r in rec:
        i = int(r["contig"])
        o = O.align_contig(contigs[i], refs[cref[i]])
        got = cigar[int(r["cigar_off"]): int(r["cigar_off"]) + int(r["n_cigar"])]
        assert (int(r["ref_start"]), int(r["ref_end"]), int(r["rev"]), int(r["mapq"])) == (o["ref_start"], o["ref_end"], o["rev"], o["mapq"])
        assert list(got) == list(o["raw"]), (i, O.cigar_str(got), O.cigar_str(o["raw"]))


def test_unalignable_contig_is_reported(ctx):
    rng = random.Random(1)
    ref = "".join(rng.choice("ACGT") for _ in range(5000)).encode()
    junk = "".join(rng.choice("ACGT") for _ in range(3000)).encode()
    rec, cigar, status = ctx.align_batch([junk, ref[100:4000]], [0, 0], [ref])
    assert status[0] == 1 and status[1] == 0 and len(rec) == 1
    assert int(rec[0]["ref_start"]) == 100 and int(rec[0]["ref_end"]) == 4000
