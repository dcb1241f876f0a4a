"""Seeded read sets for the read-signature scan (TEST INFRASTRUCTURE): the same function feeds the golden generator
(tools/make_golden_reads_scan.py, which runs the reference on them) and the test (which writes them into a BAM)."""
import random

CHROM_LEN = 2_000_000
_QOPS = (0, 1, 4, 7, 8)   # ops that consume query bases stored in SEQ


def read_sequence(d) -> str:
    n = sum(n for op, n in d["cigar"] if op in _QOPS)
    rng = random.Random(d["seq_seed"])
    return "".join(rng.choice("ACGT") for _ in range(n))


def _cigar_str(cigar):
    return "".join("%d%s" % (n, "MIDNSHP=X"[op]) for op, n in cigar)


def _random_cigar(rng, target):
    cg = []
    if rng.random() < 0.35:
        cg.append((rng.choice([4, 4, 5]), rng.randrange(1, 400)))
    left = target
    while left > 0:
        m = min(left, rng.randrange(20, 700))
        cg.append((rng.choice([0, 0, 0, 7, 8]), m))
        left -= m
        if left <= 0:
            break
        x = rng.random()
        if x < 0.30:
            cg.append((2, rng.choice([1, 5, 29, 30, 31, 60, 250, 1200])))
        elif x < 0.60:
            i = rng.choice([1, 4, 29, 30, 33, 80, 300])
            cg.append((1, i))
            left -= i
        elif x < 0.65:
            cg.append((3, rng.randrange(50, 500)))
        if rng.random() < 0.25 and cg[-1][0] in (1, 2):   # a second event close by: the merge rules
            cg.append((0, rng.choice([1, 20, 99, 100, 101, 150])))
            cg.append((cg[-2][0], rng.choice([30, 45, 10])))
    if cg[-1][0] in (1, 2, 3):
        cg.append((0, 25))
    if rng.random() < 0.35:
        cg.append((rng.choice([4, 4, 5]), rng.randrange(1, 400)))
    return cg


def _qlen(cg):
    return sum(n for op, n in cg if op in _QOPS)


def _ref_len(cg):
    return sum(n for op, n in cg if op in (0, 2, 3, 7, 8))


def make_case(seed):
    rng = random.Random(1000 + seed)
    chroms = ["chr21"] if seed != 2 else ["chr20", "chr21"]
    reads = {}
    for chrom in chroms:
        rs = []
        for j in range(220):
            pos = rng.randrange(1000, CHROM_LEN - 60000)
            if j % 11 == 0:
                pos = CHROM_LEN // 2 - rng.randrange(100, 1500)      # straddles the chunk border: seen by both tasks
            cg = _random_cigar(rng, rng.choice([300, 450, 800, 2500, 4000]))
            flag = rng.choice([0, 0, 0, 16, 16, 2048, 2064, 256, 1024])
            d = {"name": "m%d/%d/ccs" % (seed, len(rs)) + ("_DEL" if j % 50 == 7 else ""), "flag": flag, "mapq": rng.choice([0, 19, 20, 60, 60]),
                 "pos": pos, "cigar": cg, "seq_seed": rng.randrange(1 << 30), "sa": ""}
            if rng.random() < 0.45:
                d["sa"] = _make_sa(rng, chrom, d)
            rs.append(d)
        rs.sort(key=lambda d: d["pos"])
        reads[chrom] = rs
    return {"reads": reads}


def _make_sa(rng, chrom, d):
    """supplementary segments placed so that the DEL / INS / strand / other-chromosome rules all fire somewhere"""
    qlen = _qlen(d["cigar"])
    hard = sum(n for op, n in (d["cigar"][0], d["cigar"][-1]) if op == 5)
    total = qlen
    strand = "-" if d["flag"] & 16 else "+"
    end = d["pos"] + _ref_len(d["cigar"])
    out = []
    for _ in range(rng.choice([1, 1, 1, 2, 3, 8])):
        kind = rng.random()
        seg_q = rng.randrange(100, 900)
        lead = rng.randrange(0, max(1, total))
        tail = max(0, total - lead - seg_q)
        st = strand if kind < 0.7 else ("+" if strand == "-" else "-")
        ch = chrom if kind < 0.85 else "chr5"
        ref = end + rng.choice([-500, -20, 0, 10, 40, 99, 100, 101, 600, 5000, 150000]) + 1
        inner = "%dM" % seg_q if rng.random() < 0.6 else "%dM%dD%dM" % (seg_q // 2, rng.randrange(1, 80), seg_q - seg_q // 2)
        cg = ("%d%s" % (lead, rng.choice("SSH")) if lead else "") + inner + ("%d%s" % (tail, rng.choice("SSH")) if tail else "")
        out.append("%s,%d,%s,%s,%d,%d" % (ch, max(1, ref), st, cg, rng.choice([0, 10, 20, 60, 60]), rng.randrange(0, 30)))
    del hard
    return ";".join(out) + ";"
