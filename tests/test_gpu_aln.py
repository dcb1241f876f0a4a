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


def _records_by_contig(rec, cigar, n):
    by = [[] for _ in range(n)]
    for r in rec:
        cg = cigar[int(r["cigar_off"]): int(r["cigar_off"]) + int(r["n_cigar"])]
        by[int(r["contig"])].append({"ref_start": int(r["ref_start"]), "ref_end": int(r["ref_end"]), "rev": int(r["rev"]), "mapq": int(r["mapq"]),
                                     "q_start": int(r["q_start"]), "q_end": int(r["q_end"]),
                                     "cigar": [(int(x) & 0xf, int(x) >> 4) for x in cg], "raw": cg})
    return by


def _same_records(got, want, what):
    assert len(got) == len(want), (what, len(got), len(want))
    for g, w in zip(got, want):
        assert (g["ref_start"], g["ref_end"], g["rev"], g["q_start"], g["q_end"]) == (w["ref_start"], w["ref_end"], w["rev"], w["q_start"], w["q_end"]), what
        assert list(g["raw"]) == list(w["raw"]), (what, O.cigar_str(g["raw"]), O.cigar_str(w["raw"]))


def test_supplementary_records_for_svs_beyond_the_chaining_gap(ctx):
    """a 30 kb deletion / insertion with max_gap at minimap2's 20 kb breaks the chain: the rest of the contig comes back as a
    supplementary record, identical to the oracle's, and DipPAV's split-alignment rule turns the two records into the call.
    With the default max_gap (50 kb: DipPAV's max_svlen) the same SVs sit in one record's CIGAR."""
    from tests import aln_cases as A
    rng = np.random.default_rng(5)
    ref = A.rnd(rng, 150000)
    contigs = [ref[:40000] + ref[70000:], ref[:40000] + A.rnd(rng, 30000) + ref[40000:], ref[1000:90000]]
    p = ctx.default_aln_params()
    p.max_gap = 20000
    po = O.aln_default_params()
    po.max_gap = 20000
    rec, cigar, status = ctx.align_batch(contigs, [0, 0, 0], [ref], p)
    assert list(status) == [0, 0, 0]
    by = _records_by_contig(rec, cigar, 3)
    assert [len(b) for b in by] == [2, 2, 1]
    for i, c in enumerate(contigs):
        _same_records(by[i], O.align_contig_multi(c, ref, po), i)
    for i, want in ((0, ("DEL", 40000, 30000)), (1, ("INS", 40000, 30000))):
        calls = A.split_calls(by[i])
        assert len(calls) == 1 and calls[0][1] == want[0] and abs(calls[0][2] - want[1]) <= 2 and abs(calls[0][3] - want[2]) <= 2
    rec, cigar, status = ctx.align_batch(contigs, [0, 0, 0], [ref])
    assert list(status) == [0, 0, 0]
    by = _records_by_contig(rec, cigar, 3)
    assert [len(b) for b in by] == [1, 1, 1]
    for i, c in enumerate(contigs):
        _same_records(by[i], O.align_contig_multi(c, ref), i)
    assert A.events(by[0][0]) == [("DEL", 40000, 30000)] and A.events(by[1][0]) == [("INS", 40000, 30000)]


def test_duplications_beyond_max_cells(ctx):
    """duplication-type INS / DEL of 6, 8, 12 and 30 kb (exact and 2 % diverged copies), a dispersed repeat
    with variation inside a copy, a dispersed duplication, a tandem array of short units, a replacement -- each an event of more
    than max_cells cells, which used to cost the whole contig (FSV_EUNSUP).  All in one call: status 0, records bit-identical
    to the oracle's, every SV at its left-aligned position (+-1 bp) with its exact length, the contig's other SV still called."""
    from tests import aln_cases as A
    cases = A.duplication_cases() + A.other_cases()
    rec, cigar, status = ctx.align_batch([c.hap for c in cases], list(range(len(cases))), [c.ref for c in cases])
    assert (status == 0).all(), list(status)
    assert ctx.aln_stats()["n_boxes"] >= 8      # the exact copies, the array, the replacement; diverged copies keep unique seeds
    by = _records_by_contig(rec, cigar, len(cases))
    for c, got in zip(cases, by):
        A.check_case(c, got)
        _same_records(got, O.align_contig_multi(c.hap, c.ref), c.name)


def test_inversions_make_no_call(ctx):
    """a 2 kb and a 5 kb inversion (contig on either strand): the inverted piece is a record of the other
    strand, the record around it is cut in two, DipPAV's split rule makes nothing of records of different strands -- 0 calls
    beside the contig's planted deletion; records bit-identical to the oracle's"""
    from tests import aln_cases as A
    cases = A.inversion_cases()
    rec, cigar, status = ctx.align_batch([c.hap for c in cases], list(range(len(cases))), [c.ref for c in cases])
    assert (status == 0).all(), list(status)
    by = _records_by_contig(rec, cigar, len(cases))
    for c, got in zip(cases, by):
        A.check_case(c, got)
        assert A.split_calls(got) == [], c.name
        _same_records(got, O.align_contig_multi(c.hap, c.ref), c.name)


def test_left_shift_regressions_and_tandem_regions(ctx):
    """10 planted DELs in tandem-repeat regions came out 2-9 bp right of their leftmost position.  Those regions
    plus every 4th tandem-repeat region below 1500 and a run of ordinary ones, both haplotypes, through fsv_align_batch:
    CIGARs bit-identical to the oracle's, every planted SV at its left-aligned position, nothing else called."""
    from tests.test_oracle_aln import LEFT_SHIFT_REGRESSIONS, check_planted
    idx = sorted(set(LEFT_SHIFT_REGRESSIONS) | set(range(7, 1500, 32)) | set(range(2000, 2040)))
    regions = [synth.make_region(i, depth_per_hap=0.3) for i in idx]
    contigs, cref, refs = [], [], []
    for ri, r in enumerate(regions):
        refs.append(r.ref)
        for h in (0, 1):
            contigs.append(r.haps[h] if (ri + h) % 3 else synth.revcomp(r.haps[h])); cref.append(ri)
    rec, cigar, status = ctx.align_batch(contigs, cref, refs)
    assert (status == 0).all() and len(rec) == len(contigs)
    n_sv = 0
    for r in rec:
        i = int(r["contig"])
        got = cigar[int(r["cigar_off"]): int(r["cigar_off"]) + int(r["n_cigar"])]
        o = O.align_contig(contigs[i], refs[cref[i]])
        assert list(got) == list(o["raw"]), (idx[cref[i]], i & 1, O.cigar_str(got), O.cigar_str(o["raw"]))
        a = {"ref_start": int(r["ref_start"]), "cigar": [(int(x) & 0xf, int(x) >> 4) for x in got]}
        misses, extra = check_planted(regions[cref[i]], i & 1, a)
        assert not misses and extra == 0, (idx[cref[i]], i & 1, misses, extra)
        n_sv += len(a["cigar"]) // 2
    assert n_sv > 100


@pytest.mark.gpu
def test_reference_windows_with_n_runs(ctx):
    """reference windows with N (k_pack_ascii keeps a mask; an N pairs with nothing, costs 1 in a score and
    seeds as hashed sequence): records bit-identical to the oracle's, no call at the N runs, the planted SVs called -- in one batch with
    windows that have no N at all"""
    from tests import aln_cases as A
    cases = A.n_window_cases() + A.inversion_cases(sizes=(2000,))[:1] + A.duplication_cases(sizes=(6000,), divs=(0.0,))
    rec, cigar, status = ctx.align_batch([c.hap for c in cases], list(range(len(cases))), [c.ref for c in cases])
    assert (status == 0).all(), list(status)
    by = _records_by_contig(rec, cigar, len(cases))
    for c, got in zip(cases, by):
        A.check_case(c, got)
        _same_records(got, O.align_contig_multi(c.hap, c.ref), c.name)
