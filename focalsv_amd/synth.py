"""Seeded synthetic target regions: a reference window, two haplotypes
with planted DEL/INS, and HiFi-like (or ONT-like) reads delivered pre-phased exactly as
focalsv/2_phasing/output_fas.py:63-73 writes them (PS1_hp1.fa / PS1_hp2.fa, one line per read).

Region i uses seed 1000+i.  numpy's PCG64 is used instead of random.Random so that 256
regions (~400 Mbase of reads) generate in seconds; the stream is fully determined by the seed.
"""
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

_ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


@dataclass
class TruthSV:
    svtype: str   # 'DEL' | 'INS'
    pos: int      # 0-based offset of the first affected reference base (DEL) / base after which... see vcf_pos
    length: int
    gt: str       # '1/1' | '0/1'
    hap: int      # 1, 2 or 3 (both)
    seq: str = ""
    pos_left: int = -1  # the same allele left-aligned (what aligners report inside repeats)


@dataclass
class Region:
    index: int
    chrom: str
    start: int
    ref: bytes
    haps: Tuple[bytes, bytes]
    reads: Tuple[List[bytes], List[bytes]]
    truth: List[TruthSV] = field(default_factory=list)
    # what the region-cropped read BAM would say about every read: (0-based window position, BAM cigar, is_reverse)
    read_aln: Tuple[List[tuple], List[tuple]] = field(default_factory=lambda: ([], []))
    note: str = ""


def _rand_seq(rng, n):
    return _ALPHA[rng.integers(0, 4, size=n)]


def revcomp(b: bytes) -> bytes:
    return _COMP[np.frombuffer(b, dtype=np.uint8)][::-1].tobytes()


def _loguniform_int(rng, lo, hi):
    return int(round(float(np.exp(rng.uniform(np.log(lo), np.log(hi))))))


def _add_errors(rng, seq: np.ndarray, rate: float, split=(1 / 3, 1 / 3, 1 / 3)) -> np.ndarray:
    """substitution : insertion : deletion = `split` (1:1:1 unless given) at total rate `rate`."""
    n = seq.size
    if rate <= 0 or n == 0:
        return seq
    r = rng.random(n)
    if split == (1 / 3, 1 / 3, 1 / 3):      # (kept as written: the committed goldens hang on the exact comparisons)
        sub = r < rate / 3
        ins = (r >= rate / 3) & (r < 2 * rate / 3)
        dele = (r >= 2 * rate / 3) & (r < rate)
    else:
        a, b = rate * split[0], rate * (split[0] + split[1])
        sub = r < a
        ins = (r >= a) & (r < b)
        dele = (r >= b) & (r < rate)
    out = seq.copy()
    ns = int(sub.sum())
    if ns:
        out[sub] = _ALPHA[(np.searchsorted(_ALPHA, seq[sub]) + rng.integers(1, 4, size=ns)) % 4]
    keep = ~dele
    reps = np.ones(n, dtype=np.int64)
    reps[ins] = 2
    reps[~keep] = 0
    idx = np.repeat(np.arange(n), reps)
    res = out[idx]
    # the second copy of an "ins" position becomes a random base
    first = np.ones(idx.size, dtype=bool)
    first[1:] = idx[1:] != idx[:-1]
    nins = int((~first).sum())
    if nins:
        res[~first] = _ALPHA[rng.integers(0, 4, size=nins)]
    return res


def left_align(ref: np.ndarray, kind: str, pos: int, length: int, seq: np.ndarray = None):
    """VCF-style left normalisation of a planted event, as a read aligner would report it inside a repeat"""
    if kind == "DEL":
        while pos > 0 and ref[pos - 1] == ref[pos + length - 1]:
            pos -= 1
        return pos
    s = seq.copy()
    while pos > 0 and ref[pos - 1] == s[-1]:
        s = np.concatenate([ref[pos - 1:pos], s[:-1]])
        pos -= 1
    return pos


def position_tolerance(region, t) -> int:
    """how far a call may sit from the left-aligned truth position: 1 bp.  Where a haplotype-2 SNP (every 1 kb from 500, make_region)
    lies within 3 bp of a breakpoint of an SV that haplotype 2 carries, the best-scoring alignment absorbs the SNP into the gap (one
    mismatch saved, the same haplotype sequence): the call may then sit that many bases further off."""
    if not (t.hap & 2):
        return 1
    ends = (t.pos_left, t.pos_left + (t.length if t.svtype == "DEL" else 0))
    d = min(abs(e - s) for e in ends for s in range(500, len(region.ref), 1000))
    return 1 + (d if d <= 3 else 0)


def _segments(events, ref_len):
    """events [(ref_pos, 'DEL'|'INS', length)] -> [(op, ref_start, hap_start, length)] with BAM ops 0 M, 1 I, 2 D"""
    segs, r, h = [], 0, 0
    for pos, kind, n in sorted(events):
        if pos > r:
            segs.append((0, r, h, pos - r)); h += pos - r; r = pos
        if kind == "DEL":
            segs.append((2, r, h, n)); r += n
        else:
            segs.append((1, r, h, n)); h += n
    if ref_len > r:
        segs.append((0, r, h, ref_len - r))
    return segs


def _truth_cigar(segs, a, b):
    """alignment of hap[a:b) to the reference: (ref_pos, [(op, len)]); an insertion cut by the read end is soft-clipped"""
    ops, pos = [], None
    for op, rs, hs, n in segs:
        if op == 2:
            if pos is not None and hs < b and hs > a:
                ops.append((2, n))
            continue
        lo, hi = max(a, hs), min(b, hs + n)
        if hi <= lo:
            continue
        if op == 0:
            if pos is None:
                pos = rs + (lo - hs)
            ops.append((0, hi - lo))
        else:
            ops.append((1 if (pos is not None and hi < b) else 4, hi - lo))
    while ops and ops[-1][0] == 2:
        ops.pop()
    return pos, ops


def _sample_reads(rng, hap: np.ndarray, depth: float, len_lo: int, len_hi: int, err: float, min_keep: int = 3000, segs=None, aln_out=None,
                  split=(1 / 3, 1 / 3, 1 / 3)):
    reads = []
    total, target = 0, depth * hap.size
    H = hap.size
    while total < target:
        L = int(rng.integers(len_lo, len_hi + 1))
        s = int(rng.integers(-L + min_keep, H - min_keep + 1))
        a, b = max(0, s), min(H, s + L)
        if b - a < min_keep:
            continue
        seg = _add_errors(rng, hap[a:b], err, split)
        rev = rng.random() < 0.5
        if rev:
            seg = _COMP[seg][::-1]
        reads.append(seg.tobytes())
        if aln_out is not None:
            pos, ops = _truth_cigar(segs, a, b)
            aln_out.append((pos, ops, bool(rev)))
        total += b - a
    return reads


def make_region(i: int, width: int = 50_000, profile: str = "hifi", depth_per_hap: float = 15.0,
                chrom: str = "chr21", start: int = 0) -> Region:
    rng = np.random.default_rng(1000 + i)
    ref = _rand_seq(rng, width)
    edge = min(5000, width // 5)
    tandem = None
    if i % 8 == 7 and width >= 20000:
        unit = _rand_seq(rng, int(rng.integers(20, 61)))
        tpos = int(rng.integers(edge + 1000, width - edge - 3000))
        block = np.tile(unit, 2000 // unit.size + 1)[:2000]
        ref[tpos:tpos + 2000] = block
        tandem = (tpos, unit.size)

    # hap1: one DEL + one INS, >= edge from the ends and >= edge apart
    while True:
        dlen = _loguniform_int(rng, 50, 2000)
        ilen = _loguniform_int(rng, 50, 2000)
        dpos = int(rng.integers(edge, width - edge - dlen))
        ipos = int(rng.integers(edge, width - edge))
        if tandem is not None:
            # VNTR contraction: delete whole repeat units inside the block (gap placement is ambiguous there)
            units = max(1, min(dlen // tandem[1], 1500 // tandem[1]))
            while units * tandem[1] < 30:      # never below the 30 bp the reference calls (extract_contig_signature_CCS.py:351): one unit of 26-29 bases was
                units += 1
            dlen = units * tandem[1]
            dpos = tandem[0] + tandem[1] * int(rng.integers(1, max(2, (2000 - dlen) // tandem[1] - 1)))
        if abs(ipos - dpos) >= edge and abs(ipos - (dpos + dlen)) >= edge:
            if tandem is None or not (tandem[0] - 100 <= ipos <= tandem[0] + 2100):
                break
    iseq = _rand_seq(rng, ilen)
    truth = []
    homo_del = (i % 3 == 0)
    truth.append(TruthSV("DEL", dpos, dlen, "1/1" if homo_del else "0/1", 3 if homo_del else 1))
    truth.append(TruthSV("INS", ipos, ilen, "0/1", 1, iseq.tobytes().decode()))

    def apply(refarr, events):
        # events: list of (pos, kind, payload) applied right-to-left
        out = refarr
        for pos, kind, payload in sorted(events, key=lambda e: -e[0]):
            if kind == "DEL":
                out = np.concatenate([out[:pos], out[pos + payload:]])
            else:
                out = np.concatenate([out[:pos], payload, out[pos:]])
        return out

    hap1 = apply(ref, [(dpos, "DEL", dlen), (ipos, "INS", iseq)])
    ref2 = ref.copy()
    snp_pos = np.arange(500, width, 1000)
    ref2[snp_pos] = _ALPHA[(np.searchsorted(_ALPHA, ref2[snp_pos]) + 1 + (snp_pos // 1000) % 3) % 4]
    ev2 = []
    if homo_del:
        ev2.append((dpos, "DEL", dlen))
    if i % 4 == 0:
        p2 = None
        for _ in range(10000):   # narrow windows may have no room 3 kb away from both hap1 events: then hap2 gets no private INS
            c = int(rng.integers(edge, width - edge))
            if abs(c - dpos) >= 3000 and abs(c - dpos - dlen) >= 3000 and abs(c - ipos) >= 3000:
                p2 = c
                break
        if p2 is not None:
            l2 = _loguniform_int(rng, 50, 2000)
            s2 = _rand_seq(rng, l2)
            ev2.append((p2, "INS", s2))
            truth.append(TruthSV("INS", p2, l2, "0/1", 2, s2.tobytes().decode()))
    hap2 = apply(ref2, ev2)

    if profile == "hifi":
        lo, hi, err = 10_000, 20_000, 0.002
    elif profile == "ont":
        lo, hi, err = 10_000, 30_000, 0.10
    elif profile == "clr":
        # PacBio CLR-like: ~12 % error, insertions before deletions before substitutions, reads of 10-25 kb
        lo, hi, err = 10_000, 25_000, 0.12
    elif profile == "clean":
        lo, hi, err = 10_000, 20_000, 0.0
    else:
        raise ValueError(profile)
    split = (0.12, 0.55, 0.33) if profile == "clr" else (1 / 3, 1 / 3, 1 / 3)
    lo, hi = min(lo, max(1000, width // 2)), min(hi, width)
    keep = min(3000, max(500, width // 8))
    a1, a2 = [], []
    # the read records carry the events where an aligner would put them (left-aligned); the sequences are unaffected
    seg1 = _segments([(left_align(ref, "DEL", dpos, dlen), "DEL", dlen), (left_align(ref, "INS", ipos, ilen, iseq), "INS", ilen)], width)
    seg2 = _segments([(left_align(ref2, k, p, (n if k == "DEL" else len(n)), None if k == "DEL" else n), k, (n if k == "DEL" else len(n)))
                      for p, k, n in ev2], width)
    r1 = _sample_reads(rng, hap1, depth_per_hap, lo, hi, err, keep, seg1, a1, split)
    r2 = _sample_reads(rng, hap2, depth_per_hap, lo, hi, err, keep, seg2, a2, split)
    for t in truth:
        t.pos_left = left_align(ref, t.svtype, t.pos, t.length, np.frombuffer(t.seq.encode(), dtype=np.uint8) if t.svtype == "INS" else None)
    truth.sort(key=lambda t: t.pos)
    return Region(i, chrom, start, ref.tobytes(), (hap1.tobytes(), hap2.tobytes()), (r1, r2), truth, (a1, a2))


def make_repeat_region(i: int, width: int = 60_000, depth: float = 15.0, err: float = 0.002):
    """one haplotype with INTERSPERSED repeats (what the uniform-random windows of make_region lack): copies of one element --
    0.3 .. 6 kb, 2 .. 8 copies, either strand, each copy 0 .. 5 % diverged from the master (substitutions and small indels) --
    placed at random in a random window; seed 70000 + i.  -> Region with reads only in reads[0], and .note describing the layout"""
    rng = np.random.default_rng(70000 + i)
    ref = _rand_seq(rng, width)
    div = float(rng.choice([0.0, 0.005, 0.01, 0.02, 0.05]))
    kind = i % 3      # 0, 1: dispersed copies of one element; 2: many short copies (Alu-like) plus a tandem array of a long unit
    if kind < 2:
        elen = int(np.exp(rng.uniform(np.log(300), np.log(6000))))
        ncopy = int(rng.integers(2, 9))
    else:
        elen = int(rng.integers(250, 400))
        ncopy = int(rng.integers(15, 41))
    master = _rand_seq(rng, elen)
    grid = np.arange(2000, width - elen - 2000, max(elen + 200, (width - 4000) // (ncopy + 1)))
    spots = sorted(int(x) for x in rng.choice(grid, size=min(ncopy, len(grid)), replace=False))
    ncopy = len(spots)
    pieces, last = [], 0
    for sp in spots:
        copy = _add_errors(rng, master, div)
        if rng.random() < 0.5:
            copy = _COMP[copy][::-1]
        pieces += [ref[last:sp], copy]
        last = sp
    pieces.append(ref[last:])
    note2 = ""
    if kind == 2:
        unit = _rand_seq(rng, int(rng.integers(100, 500)))
        nrep = int(rng.integers(4, 16))
        pieces.append(np.concatenate([_add_errors(rng, unit, div) for _ in range(nrep)]))
        pieces.append(_rand_seq(rng, 6000))
        note2 = " + tandem %d bp x %d" % (unit.size, nrep)
    hap = np.concatenate(pieces)
    lo, hi = min(10_000, max(1000, width // 2)), min(20_000, width)
    reads = _sample_reads(rng, hap, depth, lo, hi, err, min(3000, max(500, width // 8)))
    r = Region(i, "chr21", 0, hap.tobytes(), (hap.tobytes(), b""), (reads, []), [])
    r.note = "element %d bp x %d copies, %.1f %% diverged%s" % (elen, ncopy, div * 100, note2)
    return r


def write_region_dir(region: Region, out_dir: str) -> str:
    """Region_<chr>_S<s>_E<e>/PS1_hp{1,2}.fa as output_fas.py writes them (read name line, one sequence line)."""
    import os
    d = os.path.join(out_dir, f"Region_{region.chrom}_S{region.start}_E{region.start + len(region.ref)}")
    os.makedirs(d, exist_ok=True)
    for h in (0, 1):
        with open(os.path.join(d, f"PS1_hp{h + 1}.fa"), "w") as f:
            for j, r in enumerate(region.reads[h]):
                f.write(f">r{region.index}_h{h + 1}_{j}\n{r.decode()}\n")
    return d
