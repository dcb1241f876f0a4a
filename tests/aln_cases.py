"""Contig / reference-window pairs with SVs the aligner used to refuse or miscall: duplication-type
INS / DEL of 6-30 kb (both copies fall between two unique seeds: an event of more than max_cells cells), a dispersed repeat,
a tandem array of short units, a replacement, inversions.  Shared by the oracle tests (CPU) and the -m gpu tests."""
import numpy as np

from focalsv_amd import synth

_A = np.frombuffer(b"ACGT", dtype=np.uint8)


def rnd(rng, n):
    return _A[rng.integers(0, 4, n)].tobytes()


def mutate(rng, s, rate):
    b = bytearray(s)
    for i in np.nonzero(rng.random(len(b)) < rate)[0]:
        b[i] = b"ACGT"[(b"ACGT".index(b[i]) + 1 + int(rng.integers(0, 3))) % 4]
    return bytes(b)


def left_del(ref, pos, n):
    """leftmost position of the deletion ref[pos : pos + n]"""
    while pos > 0 and ref[pos - 1] == ref[pos + n - 1]:
        pos -= 1
    return pos


def left_ins(hap, hpos, n, rpos):
    """leftmost reference position of the insertion hap[hpos : hpos + n] that sits in front of reference base rpos"""
    while hpos > 0 and hap[hpos - 1] == hap[hpos + n - 1]:
        hpos -= 1
        rpos -= 1
    return rpos


def events(rec):
    """(type, ref pos, length) of the I / D runs of at least 30 bases of one record (dict with ref_start and cigar [(op, len)])"""
    pos, out = rec["ref_start"], []
    for op, n in rec["cigar"]:
        if op == 0:
            pos += n
        elif op == 2:
            if n >= 30:
                out.append(("DEL", pos, n))
            pos += n
        elif op == 1 and n >= 30:
            out.append(("INS", pos, n))
    return out


class Case:
    def __init__(self, name, ref, hap, truth, n_rec=1, strands=None):
        self.name, self.ref, self.hap, self.truth, self.n_rec, self.strands = name, ref, hap, truth, n_rec, strands


def _with_small_del(L):
    """a 120 bp deletion at 5 000 in the left flank: the other SV of the contig that must still be called"""
    return L[:5000] + L[5120:]


def duplication_cases(sizes=(6000, 8000, 12000, 30000), divs=(0.0, 0.02)):
    out = []
    for size in sizes:
        for div in divs:
            rng = np.random.default_rng(size + int(div * 1000))
            L, R, C = rnd(rng, 12000), rnd(rng, 12000), rnd(rng, size)
            C2 = mutate(rng, C, div)
            small = ("DEL", left_del(L, 5000, 120), 120)
            # tandem duplication as an insertion: the second copy is new
            ref, hap = L + C + R, _with_small_del(L) + C + C2 + R
            at = len(L) - 120 + size
            out.append(Case("dup-INS-%d-%g" % (size, div), ref, hap, [small, ("INS", left_ins(hap, at, size, len(L) + size), size)]))
            # and the matching deletion: the reference has both copies
            ref, hap = L + C + C2 + R, _with_small_del(L) + C + R
            out.append(Case("dup-DEL-%d-%g" % (size, div), ref, hap, [small, ("DEL", left_del(ref, len(L) + size, size), size)]))
    return out


def other_cases():
    out = []
    rng = np.random.default_rng(77)
    # a dispersed two-copy repeat of 12 kb (2 % diverged) with variation inside the first copy: SNPs every 700 bases and a 300 bp
    # deletion -- the copy has no unique seeds, so its 12 k x 12 k box used to be refused
    L, M, R, C = rnd(rng, 9000), rnd(rng, 7000), rnd(rng, 9000), rnd(rng, 12000)
    C2 = mutate(rng, C, 0.02)
    ref = L + C + M + C2 + R
    hc = bytearray(C)
    for i in range(350, len(hc), 700):
        hc[i] = b"ACGT"[(b"ACGT".index(hc[i]) + 1) % 4]
    hc = bytes(hc)
    hap = _with_small_del(L) + hc[:6000] + hc[6300:] + M + C2 + R
    out.append(Case("dispersed-repeat", ref, hap, [("DEL", left_del(L, 5000, 120), 120), ("DEL", left_del(ref, len(L) + 6000, 300), 300)]))
    # a dispersed duplication: a second copy of C inserted 7 kb downstream
    ref = L + C[:8000] + M + R
    hap = _with_small_del(L) + C[:8000] + M + C[:8000] + R
    at = len(L) - 120 + 8000 + len(M)
    out.append(Case("dispersed-dup-INS", ref, hap, [("DEL", left_del(L, 5000, 120), 120), ("INS", left_ins(hap, at, 8000, len(L) + 8000 + len(M)), 8000)]))
    # a tandem array of 50-base units, 130 copies against 260: nothing in it can be seeded, the box is closed from its corners
    unit = rnd(rng, 50)
    ref = L + unit * 130 + R
    hap = _with_small_del(L) + unit * 260 + R
    out.append(Case("array-INS", ref, hap, [("DEL", left_del(L, 5000, 120), 120), ("INS", left_ins(hap, len(L) - 120 + 6500, 6500, len(L) + 6500), 6500)]))
    # 8 kb replaced by 10 kb of unrelated sequence
    X, Y = rnd(rng, 8000), rnd(rng, 10000)
    ref = L + X + R
    hap = _with_small_del(L) + Y + R
    out.append(Case("replacement", ref, hap, [("DEL", left_del(L, 5000, 120), 120), ("INS", len(L), 10000), ("DEL", len(L), 8000)]))
    return out


def inversion_cases(sizes=(2000, 5000)):
    out = []
    for size in sizes:
        for rc in (0, 1):
            rng = np.random.default_rng(size)
            L, M, R = rnd(rng, 12000), rnd(rng, size), rnd(rng, 15000)
            ref = L + M + R
            hap = _with_small_del(L) + synth.revcomp(M) + R
            if rc:
                hap = synth.revcomp(hap)
            out.append(Case("inversion-%d-%s" % (size, "-+"[1 - rc]), ref, hap, [("DEL", left_del(L, 5000, 120), 120)], n_rec=3,
                            strands=[rc, rc, 1 - rc]))
    return out


def n_window_cases():
    """reference windows that hold runs of N (they used to be stored -- and to seed and match -- as poly-A).
    The contig carries real sequence where the reference has N; minimap2 scores an N against anything as -1 and skips k-mers with
    an N: the alignment runs through a short run as 'M', nothing is called there, and the contig's real SVs still are."""
    out = []
    rng = np.random.default_rng(4242)
    L, M, R = rnd(rng, 12000), rnd(rng, 9000), rnd(rng, 12000)
    small = ("DEL", left_del(L, 5000, 120), 120)
    hap = _with_small_del(L) + M + R
    for name, runs in (("N-run-3", [(3000, 3)]), ("N-run-200", [(3000, 200)]), ("N-run-1500", [(3000, 1500)]),
                       ("N-runs-many", [(500, 1), (1200, 40), (4000, 300), (7000, 17)])):
        m = bytearray(M)
        for at, n in runs:
            m[at:at + n] = b"N" * n
        out.append(Case(name, L + bytes(m) + R, hap, [small]))
    # an N run of 60 next to a real 400 bp deletion (30 bases apart) and one inside a 500 bp insertion's flank
    m = bytearray(M)
    m[2970:3030 - 30] = b"N" * 30
    ref = L + bytes(m) + R
    hap2 = _with_small_del(L) + M[:3100] + M[3500:] + R
    out.append(Case("N-run-beside-DEL", ref, hap2, [small, ("DEL", left_del(L + M + R, len(L) + 3100, 400), 400)]))
    ins = rnd(rng, 500)
    m = bytearray(M)
    m[5050:5090] = b"N" * 40
    ref = L + bytes(m) + R
    hap3 = _with_small_del(L) + M[:5000] + ins + M[5000:] + R
    out.append(Case("N-run-beside-INS", ref, hap3, [small, ("INS", left_ins(hap3, len(L) - 120 + 5000, 500, len(L) + 5000), 500)]))
    # a window that starts and ends in N, and poly-A in the contig opposite an N run (what "N stored as A" made an exact match)
    m = bytearray(M)
    m[3000:3300] = b"N" * 300
    hm = bytearray(M)
    hm[3000:3300] = b"A" * 300
    out.append(Case("N-run-vs-polyA", b"N" * 700 + L[700:] + bytes(m) + R[:-900] + b"N" * 900, _with_small_del(L) + bytes(hm) + R, [small]))
    return out


def check_case(case, recs):
    """recs: the records of the contig (dicts with ref_start, rev, cigar): every planted SV within 1 bp with its exact length,
    nothing else; the expected number of records and their strands"""
    assert len(recs) == case.n_rec, (case.name, len(recs))
    if case.strands is not None:
        assert [r["rev"] for r in recs] == case.strands, (case.name, [r["rev"] for r in recs])
    ev = [e for r in recs for e in events(r)]
    miss = [t for t in case.truth if not any(k == t[0] and n == t[2] and abs(p - t[1]) <= 1 for k, p, n in ev)]
    assert not miss and len(ev) == len(case.truth), (case.name, ev, case.truth)
    for r in recs:
        assert sum(n for op, n in r["cigar"] if op in (0, 1, 4)) == len(case.hap), case.name


def split_calls(recs):
    """what DipPAV's split rule makes of consecutive records (sorted by position, as in the BAM)"""
    from focalsv_amd.dippav import signatures as S
    segs = [S.AlignedSegment("chr21", r["ref_start"], r["ref_end"], r["cigar"], "contig_hp1_0", bool(r["rev"]), 60, None) for r in recs]
    segs.sort(key=lambda x: x.pos)
    out = []
    for a, b in zip(segs, segs[1:]):
        d, i = S.extract_sig_from_split(a, b)
        out += d + i
    return out
