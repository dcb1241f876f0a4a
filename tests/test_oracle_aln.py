"""oracle/aln.c: (1) the DP against the reference tree's own ksw2 (golden from ksw_extz2_sse, single affine),
(2) contig-vs-window alignment recovers the planted SVs."""
import json
import os

from focalsv_amd import synth
from tests import oracle_lib as O


def test_nw_matches_ksw2_golden(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "ksw_extz2.json")))["cases"]
    checked = 0
    for c in cases:
        p = O.aln_default_params()
        p.a, p.b, p.q, p.e, p.q2, p.e2 = c["a"], c["b"], c["q"], c["e"], -1, -1
        import re
        ops = re.findall(r"(\d+)([MID])", c["cigar"])
        qlen = sum(int(n) for n, o in ops if o in "MI")
        tlen = sum(int(n) for n, o in ops if o in "MD")
        if qlen != len(c["query"]) or tlen != len(c["target"]):
            continue  # z-dropped extension, not a global alignment
        sc, cg = O.nw(c["target"].encode(), c["query"].encode(), p)
        assert sc == c["score"], c
        assert O.cigar_str(cg) == c["cigar"], (O.cigar_str(cg), c["cigar"])
        checked += 1
    assert checked >= 40


def _events(aln):
    pos, out = aln["ref_start"], []
    for op, n in aln["cigar"]:
        if op == 0:
            pos += n
        elif op == 2:
            if n >= 30:
                out.append(("DEL", pos, n))
            pos += n
        elif op == 1 and n >= 30:
            out.append(("INS", pos, n))
    return out


def test_alignment_recovers_planted_svs():
    for i in (0, 1, 4, 7, 15):
        r = synth.make_region(i)
        for h in (0, 1):
            for contig in (r.haps[h], synth.revcomp(r.haps[h])):
                a = O.align_contig(contig, r.ref)
                assert a is not None and a["mapq"] == 60
                assert a["ref_start"] == 0 and a["ref_end"] == len(r.ref)
                qlen = sum(n for op, n in a["cigar"] if op in (0, 1, 4))
                assert qlen == len(contig)
                ev = _events(a)
                truth = [(t.svtype, t.pos, t.length) for t in r.truth if t.hap in (h + 1, 3)]
                assert len(ev) == len(truth), (i, h, ev, truth)
                for (ty, pos, ln), (tty, tpos, tln) in zip(sorted(ev, key=lambda e: e[1]), truth):
                    assert ty == tty and ln == tln, (i, h, ev, truth)
                    # our gaps are left-aligned; the planted position can only be at or right of it (repeat context)
                    assert pos <= tpos, (i, h, ev, truth)
                    if i % 8 != 7:
                        assert tpos - pos <= 8, (i, h, ev, truth)
