"""Seeded inputs for the read-based draft caller (TEST INFRASTRUCTURE): synthetic regions' reads with their true alignments, the
signature files the (reference-pinned) scan gives for them, and the reference sequence.  Shared by the golden generator
(tools/make_golden_reads_cluster.py, which runs the reference's clustering on them) and the test."""
import random

from focalsv_amd import reads_scan, synth

CHROM_LEN = 1_000_000


def make_case(seed, n_regions=4, width=30000, chroms=("chr21",)):
    rng = random.Random(seed)
    reads, out, ref = {}, {'DEL': {}, 'INS': {}}, {}
    for ci, chrom in enumerate(chroms):
        seq = [rng.choice("ACGT") for _ in range(CHROM_LEN)]
        rs = []
        for k in range(n_regions):
            r = synth.make_region(seed * 100 + ci * 10 + k, width=width, chrom=chrom, start=100000 + k * (width + 25000))
            seq[r.start:r.start + len(r.ref)] = r.ref.decode()
            for h in (0, 1):
                for j, (pos, ops, rev) in enumerate(r.read_aln[h]):
                    need = sum(n for op, n in ops if op in (0, 1, 4))
                    bases = r.reads[h][j]
                    bases = ((synth.revcomp(bases) if rev else bases) + b"A" * need)[:need].decode()
                    flag = 16 if rev else 0
                    if rng.random() < 0.08:
                        flag |= rng.choice([256, 1024, 2048])      # not counted as primary by the genotyper
                    name = "r%d_h%d_%d" % (r.index, h + 1, j)
                    end = r.start + pos + sum(n for op, n in ops if op in (0, 2))
                    rs.append({"name": name, "pos": r.start + pos, "end": end, "flag": flag, "cigar": [list(o) for o in ops], "seq": bases})
                    reads_scan.scan_record(chrom, r.start + pos, end, flag, 60, [tuple(o) for o in ops], len(bases), name, bases, '', out)
        rs.sort(key=lambda d: d["pos"])
        reads[chrom] = rs
        ref[chrom] = "".join(seq)
    lines = reads_scan.format_lines(out)
    return {"reads": reads, "ref": ref, "del_sigs": "".join(reads_scan.sort_sigs(lines, "DEL")), "ins_sigs": "".join(reads_scan.sort_sigs(lines, "INS"))}
