#!/usr/bin/env python3
"""Mint golden vectors for the host-side SV logic by importing the reference's own modules.

Runs only in the build container: imports focalsv/4_sv_calling/Dippav/*.py from /root/reference with
`pysam` and `edlib` (absent third-party packages) replaced by empty stand-in modules -- the functions exercised
here never touch pysam; remove_redundancy's edit_sim is given a plain Levenshtein distance (what edlib computes).
sys.dont_write_bytecode keeps the read-only tree clean.  Only inputs and outputs are written (tests/golden/dippav_*.json).
"""
import json
import os
import random
import sys
import tempfile
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/focalsv/4_sv_calling/Dippav"
sys.path.insert(0, REF)
sys.modules['pysam'] = types.ModuleType('pysam')
_ed = types.ModuleType('edlib')


def _lev(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


_ed.align = lambda a, b: {"editDistance": _lev(a, b)}
sys.modules['edlib'] = _ed

import extract_contig_signature_CCS as CCS  # noqa: E402
import extract_contig_signature_CLR as CLR  # noqa: E402
import extract_contig_signature_ONT as ONT  # noqa: E402
import FP_filter_v1 as FP  # noqa: E402
import remove_redundancy as RR  # noqa: E402
import extract_reads_signature as RS  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class Rec:
    def __init__(self, d):
        self.__dict__.update(d)


def rand_cigar(rng, clip_head=None, clip_tail=None, hard=False):
    ops = []
    if clip_head:
        ops.append([5 if hard else 4, clip_head])
    n = rng.randint(1, 9)
    for i in range(n):
        ops.append([0, rng.randint(5, 900)])
        if i + 1 < n:
            kind = rng.choice([1, 2])
            ln = rng.choice([rng.randint(1, 29), rng.randint(30, 120), rng.randint(100, 400), rng.randint(250, 2500)])
            ops.append([kind, ln])
            if rng.random() < 0.4:  # a close second event, to exercise the per-contig merges
                ops.append([0, rng.randint(1, 300)])
                ops.append([kind, rng.choice([rng.randint(101, 200), rng.randint(260, 330), rng.randint(330, 900)])])
    ops.append([0, rng.randint(5, 500)])
    if clip_tail:
        ops.append([5 if hard else 4, clip_tail])
    return ops


def rec_from(rng, name, pos, cigar, rev=None, mapq=None):
    ref_len = sum(n for op, n in cigar if op in (0, 2))
    return {"reference_name": "chr21", "pos": pos, "reference_end": pos + ref_len, "cigar": cigar, "qname": name,
            "is_reverse": bool(rng.random() < 0.5) if rev is None else rev, "mapq": rng.choice([60, 60, 60, 50, 20]) if mapq is None else mapq,
            "seq": None}


def main():
    rng = random.Random(20261004)
    # --- extract_sig_from_cigar (CCS flavour is shared by CLR/ONT) + reads flavour
    cig = []
    for i in range(120):
        r = rec_from(rng, "contig_hp1_%d" % i, rng.randint(0, 10 ** 6), rand_cigar(rng, rng.choice([None, None, 37]), rng.choice([None, 55]), rng.random() < 0.3))
        d, ins, ref_end, ctg = CCS.extract_sig_from_cigar(Rec(r), 30)
        d2, ins2, _, _ = RS.extract_sig_from_cigar(Rec(r), 30)
        cig.append({"rec": r, "del": d, "ins": ins, "ref_end": ref_end, "ctg": ctg, "reads_del": d2, "reads_ins": ins2,
                    "start_end": list(CCS.get_read_start_end(r["cigar"])),
                    "clr_ins_pct": CLR.ins_pct(r["cigar"]), "clr_var_dist": CLR.var_dist(r["cigar"])})
    # --- extract_sig_from_split for the three data types
    spl = []
    for i in range(400):
        total = rng.randint(3000, 60000)
        cut1 = rng.randint(500, total - 500)
        cut2 = cut1 + rng.choice([0, 0, rng.randint(-400, 400), rng.randint(30, 5000), -rng.randint(30, 3000)])
        cut2 = min(max(cut2, 1), total - 1)
        c1 = [[0, cut1], [rng.choice([4, 5]), total - cut1]]
        c2 = [[rng.choice([4, 5]), cut2], [0, total - cut2]]
        p1 = rng.randint(1000, 10 ** 6)
        gap = rng.choice([0, rng.randint(-2500, 400), rng.randint(30, 6000), rng.randint(-3500, -30), 60000])
        p2 = max(p1, p1 + cut1 + gap)
        rev = rng.random() < 0.5
        mq = rng.choice([60, 60, 60, 40])
        r1 = rec_from(rng, "contig_hp1_7", p1, c1, rev, mq)
        r2 = rec_from(rng, "contig_hp1_7", p2, c2, rev if rng.random() < 0.9 else not rev, 60)
        item = {"r1": r1, "r2": r2}
        for nm, mod in (("CCS", CCS), ("CLR", CLR), ("ONT", ONT)):
            d, ins = mod.extract_sig_from_split(Rec(r1), Rec(r2), 50, 50000)
            item[nm] = {"del": d, "ins": ins}
        spl.append(item)
    # --- clustering / merge_all / pair_sig
    def rand_sigs(n, svtype, hp, src="cigar"):
        sigs, pos = [], rng.randint(1000, 5000)
        for i in range(n):
            pos += rng.choice([0, rng.randint(1, 60), rng.randint(50, 250), rng.randint(500, 5000)])
            ln = rng.choice([rng.randint(30, 80), rng.randint(50, 600), rng.randint(300, 3000)])
            cs = rng.randint(0, 40000)
            sigs.append(["chr21", svtype, pos, ln, "contig_%s_%d" % (hp, rng.randint(0, 3)), cs, cs + (ln if svtype == 'INS' else 1),
                         rng.choice("+-"), src, 60 if src == "cigar" else "60-60"])
        return sigs
    clu = []
    for i in range(60):
        d = CCS.sort_sig(rand_sigs(rng.randint(0, 14), 'DEL', 'hp1'))
        ins = CCS.sort_sig(rand_sigs(rng.randint(0, 14), 'INS', 'hp1'))
        ds = CCS.sort_sig(rand_sigs(rng.randint(0, 5), 'DEL', 'hp1', "split-alignment"))
        iss = CCS.sort_sig(rand_sigs(rng.randint(0, 5), 'INS', 'hp1', "split-alignment"))
        item = {"del": d, "ins": ins, "del_split": ds, "ins_split": iss,
                "cluster_del": CCS.cluster_del(d, 100, 0.5, 0.5) if d else [], "cluster_ins": CCS.cluster_ins(ins, 100, 0.5) if ins else []}
        item["merge_all"] = CCS.merge_all(item["cluster_del"], item["cluster_ins"],
                                          CCS.cluster_del(ds) if ds else [], CCS.cluster_ins(iss) if iss else [])
        clu.append(item)
    pair = []
    for i in range(60):
        h1 = CCS.sort_sig(rand_sigs(rng.randint(0, 8), 'DEL', 'hp1') + rand_sigs(rng.randint(0, 8), 'INS', 'hp1'))
        h2 = []
        for s in h1:
            if rng.random() < 0.6:
                t = list(s)
                t[2] += rng.choice([0, 3, -20, 150, 260])
                t[3] = max(30, int(t[3] * rng.choice([1.0, 0.9, 1.3, 0.4])))
                t[4] = t[4].replace("hp1", "hp2")
                h2.append(t)
        h2 = CCS.sort_sig(h2 + rand_sigs(rng.randint(0, 3), 'INS', 'hp2'))
        pair.append({"hp1": h1, "hp2": h2, "paired": CCS.pair_sig([list(s) for s in h1], [list(s) for s in h2], 1000, 200, 0.5, 0.5)})
    # --- write_vcf
    ref_seq = "".join(rng.choice("ACGT") for _ in range(60000))
    contigs = {"contig_hp%d_%d" % (h, i): "".join(rng.choice("acgtACGT") for _ in range(45000)) for h in (1, 2) for i in range(4)}
    vcfs = []
    for i in range(6):
        h1 = CCS.sort_sig(rand_sigs(6, 'DEL', 'hp1') + rand_sigs(6, 'INS', 'hp1'))
        if i == 0:
            h1[0][5] = 0; h1[0][6] = h1[0][3] if h1[0][1] == 'INS' else 1; h1[0][7] = '-'  # the `-0` slice quirk
        h2 = CCS.sort_sig(rand_sigs(3, 'DEL', 'hp2') + rand_sigs(3, 'INS', 'hp2'))
        paired = CCS.pair_sig([list(s) for s in h1], [list(s) for s in h2], 1000, 200, 0.5, 0.5)
        CCS.ref_seq = ref_seq
        CCS.dc_contig = contigs
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "o.vcf")
            CCS.write_vcf([list(s) for s in paired], out, None, None, os.path.join(REF, "header"))
            vcfs.append({"paired": paired, "vcf": open(out).read()})
    # --- FP filter
    fps = []
    for i in range(40):
        sigs = [s[:4] for s in CCS.sort_sig(rand_sigs(12, rng.choice(['DEL', 'INS']), 'hp1'))]
        reads = [s[:4] for s in CCS.sort_sig(rand_sigs(40, 'DEL', 'r') + rand_sigs(40, 'INS', 'r'))]
        fps.append({"sigs": sigs, "reads": reads, "support": FP.eval_sig(sigs, reads, 1000, 250, 500, 0.5),
                    "support_default": FP.eval_sig(sigs, reads, 1000)})
    # --- remove_redundancy end to end on the VCF text produced above (ties in length avoided: they are hash-seed dependent upstream)
    reds = []
    for i, v in enumerate(vcfs[1:]):  # vcf 0 carries the zero-length `-0` INS, on which the reference itself divides by zero
        lines = v["vcf"].splitlines(True)
        body = [l for l in lines if l[0] != '#']
        extra = []
        for l in body[:6]:  # add near-duplicates so that clusters exist
            d = l.split('\t')
            d[1] = str(int(d[1]) + rng.randint(1, 80)); d[2] = d[2] + "x"
            if len(d[4]) > len(d[3]):
                d[4] = d[4][: max(2, int(len(d[4]) * 0.9))]
            else:
                d[3] = d[3][: max(2, int(len(d[3]) * 0.8))]
            extra.append('\t'.join(d))
        text = "".join([l for l in lines if l[0] == '#'] + body + extra)
        with tempfile.TemporaryDirectory() as tmp:
            inp = os.path.join(tmp, "in.vcf")
            open(inp, "w").write(text)
            RR.remove_redundancy(inp, os.path.join(tmp, "out"))
            reds.append({"vcf": text, "kept": open(os.path.join(tmp, "out", "dippav_variant_no_redundancy.vcf")).read(),
                         "dropped": open(os.path.join(tmp, "out", "dippav_variant_redundancy.vcf")).read()})
    sims = []
    for i in range(40):
        a = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 300)))
        b = list(a)
        for _ in range(rng.randint(0, 40)):
            p = rng.randrange(len(b) + 1)
            b[p:p + rng.randint(0, 2)] = [rng.choice("ACGT")] * rng.randint(0, 2)
        b = "".join(b) or "A"
        sims.append({"a": a, "b": b, "sim": RR.edit_sim(a, b), "dist": _lev(a, b)})
    json.dump({"source": "focalsv/4_sv_calling/Dippav/*.py imported from /root/reference (pysam/edlib stubbed)",
               "cigar": cig, "split": spl, "cluster": clu, "pair": pair}, open(os.path.join(OUT, "dippav_signatures.json"), "w"), separators=(",", ":"))
    json.dump({"source": "write_vcf / eval_sig / remove_redundancy of the reference", "ref_seq": ref_seq, "contigs": contigs, "vcf": vcfs,
               "fp": fps, "redundancy": reds, "edit_sim": sims}, open(os.path.join(OUT, "dippav_vcf.json"), "w"), separators=(",", ":"))
    print("ok", len(cig), len(spl), len(clu), len(pair), len(vcfs), len(fps), len(reds))


def main_reads():
    """extract_reads_signature.py end to end (cigar source + split source + merge) on synthetic records, through a stand-in for
    pysam.AlignmentFile -> tests/golden/dippav_reads_sig.json (separate RNG: the other golden files stay as they are)"""
    import tempfile
    rng = random.Random(77)
    cases = []
    for ci in range(12):
        recs = []
        pos = 1000
        for i in range(rng.randint(20, 60)):
            pos += rng.randint(0, 4000)
            name = "read_%d" % i
            if rng.random() < 0.25:   # a split read: two or three records of one name, in position order
                total = rng.randint(3000, 30000)
                cut1 = rng.randint(500, total - 500)
                cut2 = min(max(cut1 + rng.choice([0, rng.randint(-200, 200), rng.randint(30, 4000), -rng.randint(30, 2000)]), 1), total - 1)
                rev = rng.random() < 0.5
                mq = rng.choice([60, 60, 30, 0])
                r1 = rec_from(rng, name, pos, [[0, cut1], [rng.choice([4, 5]), total - cut1]], rev, mq)
                gap = rng.choice([0, rng.randint(-20, 25), rng.randint(30, 6000), rng.randint(-3000, -30), 70000])
                r2 = rec_from(rng, name, max(pos, pos + cut1 + gap), [[rng.choice([4, 5]), cut2], [0, total - cut2]], rev if rng.random() < 0.9 else not rev, rng.choice([60, 10]))
                recs += [r1, r2]
                if rng.random() < 0.2:
                    recs.append(rec_from(rng, name, r2["reference_end"] + rng.randint(0, 500), [[4, total - 300], [0, 300]], rev, 60))
            else:
                recs.append(rec_from(rng, name, pos, rand_cigar(rng, rng.choice([None, 37]), rng.choice([None, 55]), rng.random() < 0.3)))
        recs.sort(key=lambda r: r["pos"])   # fetch() yields position order

        class FakeBam:
            def __init__(self, path):
                pass

            def fetch(self, chrom):
                return [Rec(r) for r in recs]
        RS.pysam.AlignmentFile = FakeBam
        with tempfile.TemporaryDirectory() as tmp:
            RS.extract_reads_signature(21, "none.bam", tmp)
            lines = open(os.path.join(tmp, "reads_signature", "chr21_reads_sig.txt")).read().splitlines()
        cases.append({"records": recs, "reads_sig_lines": lines})
    json.dump({"source": "extract_reads_signature.extract_reads_signature of the reference on synthetic records", "cases": cases},
              open(os.path.join(OUT, "dippav_reads_sig.json"), "w"), separators=(",", ":"))
    print("reads ok", len(cases), sum(len(c["reads_sig_lines"]) for c in cases))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "reads":
        main_reads()
    else:
        main()
