#!/usr/bin/env python3
"""Robustness sweep on the GPU: unusual read sets through assemble + align + call, every set compared with the oracle where
the oracle is fast enough.  Every case has an expected outcome (all planted SVs called within 1 bp, no false call, contigs equal
to the oracle's -- or, for the degenerate read sets, exactly what is stated at the call site); the exit code is non-zero when
one differs.  Not part of the test suite (minutes); run under `timeout`."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from focalsv_amd import _lib, pipeline, synth
from tests import oracle_lib as O

FAILED = []

def run(ctx, name, regions, check=True, expect=None):
    """expect: None = every planted SV called, nothing else; or the (tp, fp, fn) the case must give"""
    t = time.time()
    batch = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r) for r in regions])
    try:
        res = pipeline.run_hot_path(ctx, batch)
    finally:
        batch.free(ctx)
    ok = True
    if check:
        for ri, r in enumerate(regions):
            for h in (0, 1):
                oc, _ = O.assemble(r.reads[h])
                mine = [c for (rg, hp, c) in res.contigs if rg == ri and hp == h + 1]
                if mine != oc:
                    ok = False
                    print("   MISMATCH", name, ri, h, [len(c) for c in mine], [len(c) for c in oc])
    calls = pipeline.parse_calls(res.lines)
    truth = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for r in regions for t in r.truth]
    tp, fp, fn, gt = pipeline.match_truth(calls, truth, bp_tol=1, len_tol=0.02)
    print("%-34s sets %3d status %s contigs %3d calls tp/fp/fn %d/%d/%d  oracle %s  %.1fs" % (
        name, 2 * len(regions), sorted(set(int(s) for s in res.set_status)), len(res.contigs), tp, fp, fn, "same" if ok else "DIFF", time.time() - t), flush=True)
    want = (len(truth), 0, 0) if expect is None else expect
    if (tp, fp, fn) != want or not ok:
        FAILED.append((name, (tp, fp, fn), want, ok))

def main():
    with _lib.Context(0) as ctx:
        run(ctx, "wide 200 kb, 15x", [synth.make_region(700, width=200000, start=0)], check=False)
        run(ctx, "deep 40x, 50 kb", [synth.make_region(701, depth_per_hap=40.0, start=300000)])
        run(ctx, "deep 80x, 26 kb", [synth.make_region(702, width=26000, depth_per_hap=80.0, start=600000)])
        run(ctx, "many small 14 kb x 64", [synth.make_region(710 + i, width=14000 + 30000, start=i * 60000) for i in range(64)], check=False)
        run(ctx, "clean reads (0 error)", [synth.make_region(703, profile="clean", start=0)])
        run(ctx, "thin 5x", [synth.make_region(704, depth_per_hap=5.0, start=0)])
        run(ctx, "mixed widths", [synth.make_region(720 + i, width=w, start=i * 300000) for i, w in enumerate([20000, 35000, 80000, 120000, 50000, 44000])], check=False,
            expect=(13, 0, 1))   # one haplotype of the 20 kb window lays out as a chain of < 4 reads: hifiasm cuts it as a tip and writes no contig, so do we (FSV_W_NO_LAYOUT)
        # reads with N and lower-case bases
        r = synth.make_region(705, start=0)
        rd = list(r.reads[0]); rd[0] = rd[0][:100] + b"N" * 20 + rd[0][120:]; rd[1] = rd[1].lower()
        r.reads = (rd, r.reads[1])
        run(ctx, "N run and lower case in reads", [r], check=False)
        # read lengths far from the bench's: 40-80 kb reads over a 200 kb window, and a set of very short reads
        import numpy as np
        rng = np.random.default_rng(9)
        big = synth.make_region(706, width=200000, start=0)
        hap = np.frombuffer(big.haps[0], dtype=np.uint8)
        long_reads = []
        for _ in range(60):
            L = int(rng.integers(40000, 80000)); s0 = int(rng.integers(0, hap.size - L))
            long_reads.append(synth._add_errors(rng, hap[s0:s0 + L], 0.002).tobytes())
        big.reads = (long_reads, big.reads[1])
        run(ctx, "40-80 kb reads", [big])
        tiny = synth.make_region(707, start=0)
        tiny.reads = ([rd[:300] for rd in tiny.reads[0]], [rd[:40] for rd in tiny.reads[1]])
        run(ctx, "300-base and 40-base reads", [tiny], expect=(0, 0, 2))   # nothing to assemble: no contig, no call, no crash
        one = synth.make_region(708, start=0)
        one.reads = (one.reads[0][:1], [])
        run(ctx, "one read / empty set", [one], expect=(0, 0, 3))
    for f in FAILED:
        print("FAILED", f)
    print("stress done:", "all as expected" if not FAILED else "%d case(s) off" % len(FAILED))
    return 1 if FAILED else 0

sys.exit(main())
