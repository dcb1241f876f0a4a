"""End to end on the GPU: read sets -> contigs -> alignments -> VCF, checked against the planted truth and against
the same host logic fed with the CPU oracle's contigs and alignments."""
import numpy as np
import pytest

from focalsv_amd import _lib, pipeline, synth
from focalsv_amd.dippav import signatures as S
from focalsv_amd.dippav.variant_call import WindowedRef, call_chromosome
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    with _lib.Context(0) as c:
        yield c


def _regions(ids):
    return [synth.make_region(i, start=i * 60000) for i in ids]


def test_calls_match_truth_and_oracle_path(ctx):
    rs = _regions([0, 3, 7, 12])
    batch = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r) for r in rs])
    try:
        res = pipeline.run_hot_path(ctx, batch)
    finally:
        batch.free(ctx)
    calls = pipeline.parse_calls(res.lines)
    truth = [(r.chrom, t.svtype, r.start + t.pos, t.length, t.gt) for r in rs for t in r.truth]
    tp, fp, fn, gt_ok = pipeline.match_truth(calls, truth, bp_tol=1, len_tol=0.02, left_shift_ok=2000)
    assert (tp, fp, fn) == (len(truth), 0, 0), (calls, truth)
    assert gt_ok == tp
    # +-1 bp / exact SVLEN against the left-aligned truth (the tolerance north_star states), tandem-repeat region included
    truth_left = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for r in rs for t in r.truth]
    tp1, fp1, fn1, _ = pipeline.match_truth(calls, truth_left, bp_tol=1, len_tol=0.0, left_shift_ok=0)
    assert (tp1, fp1, fn1) == (len(truth), 0, 0), (calls, truth_left)
    # CPU reference path: oracle contigs + oracle alignments through the same host logic -> identical VCF body
    names, recs, contig_seq, cnt = [], [], {}, {1: 0, 2: 0}
    for r in rs:
        for h in (0, 1):
            for c in O.assemble(r.reads[h])[0]:
                name = "contig_hp%d_%d" % (h + 1, cnt[h + 1]); cnt[h + 1] += 1
                a = O.align_contig(c, r.ref)
                contig_seq[name] = c.decode()
                recs.append(S.AlignedSegment(r.chrom, r.start + a["ref_start"], r.start + a["ref_end"], a["cigar"], name, bool(a["rev"]), a["mapq"], None))
    recs.sort(key=lambda x: x.pos)
    ref = WindowedRef()
    for r in rs:
        ref.add(r.start, r.ref.decode())
    _, body = call_chromosome(recs, "chr21", ref, contig_seq, 'CCS')
    assert body == res.raw_lines


def test_real_bed_geometry_config3(ctx, golden_dir):
    """BASELINE.json configs[2] geometry: region widths of the reference's chr21 auto-mode BED (14 kb .. >100 kb), synthetic reads
    laid over them; every planted SV must come back within 1 bp of the left-aligned truth."""
    import json
    import os
    bed = json.load(open(os.path.join(golden_dir, "bed_chr21_regions.json")))["chr21"]
    picks = sorted(bed, key=lambda r: r[1] - r[0])
    chosen = [picks[0], picks[len(picks) // 2], picks[-8], picks[len(picks) // 3]]  # smallest, median, a wide one (> 60 kb)
    # `samtools view bam chr:start-end` (1_crop_bam.py:74) returns whole reads that touch the region, so a read set spans about a
    # read length beyond either end of its BED line: the synthetic haplotype is the region +- 15 kb
    rs = [synth.make_region(300 + i, width=b - a + 30000, start=max(0, a - 15000)) for i, (a, b) in enumerate(chosen)]
    batch = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r) for r in rs])
    try:
        res = pipeline.run_hot_path(ctx, batch)
    finally:
        batch.free(ctx)
    assert (res.set_status == 0).all() and (res.contig_status == 0).all()
    calls = pipeline.parse_calls(res.lines)
    truth = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for r in rs for t in r.truth]
    tp, fp, fn, gt_ok = pipeline.match_truth(calls, truth, bp_tol=1, len_tol=0.02)
    assert (tp, fp, fn) == (len(truth), 0, 0), (calls, truth)


def test_unphased_reads_in_memory_path(ctx):
    """a region whose reads are all unphased (both haplotypes in one set) next to a phased region in the same batch: the
    partition inside fsv_assemble_batch recovers both haplotypes and every planted SV comes out with its genotype"""
    ra, rb, rc = (synth.make_region(i, start=i * 60000) for i in (8, 9, 12))
    inputs = [pipeline.region_from_synth(r) for r in (ra, rb, rc)]
    un = inputs[0]
    un.reads_unphased, un.reads_hp1, un.reads_hp2 = un.reads_hp1 + un.reads_hp2, [], []
    hom = inputs[2]            # only one haplotype's reads, unphased: one contig that stands for both haplotypes
    hom.reads_unphased, hom.reads_hp1, hom.reads_hp2 = list(hom.reads_hp1), [], []
    batch = pipeline.upload_regions(ctx, inputs)
    try:
        res = pipeline.run_hot_path(ctx, batch)
    finally:
        batch.free(ctx)
    assert (res.contig_status == 0).all()
    calls = pipeline.parse_calls(res.raw_lines)
    truth = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for r in (ra, rb) for t in r.truth]
    truth += [(rc.chrom, t.svtype, rc.start + t.pos_left, t.length, "1/1") for t in rc.truth if t.hap in (1, 3)]
    tp, fp, fn, gt_ok = pipeline.match_truth(calls, truth, bp_tol=1, len_tol=0.02)
    assert (tp, fp, fn, gt_ok) == (len(truth), 0, 0, len(truth)), (calls, truth)


def test_two_lanes_give_the_same_calls():
    """pipeline.run_hot_path_lanes: halves of the batch on two contexts / streams / host threads -> the single-lane VCF body"""
    rs = [synth.make_region(i, start=i * 60000) for i in range(20, 28)]
    inputs = [pipeline.region_from_synth(r) for r in rs]
    with _lib.Context(0) as c0:
        b0 = pipeline.upload_regions(c0, inputs)
        one = pipeline.run_hot_path(c0, b0)
        b0.free(c0)
    ctxs = [_lib.Context(0), _lib.Context(0)]
    try:
        batches = [pipeline.upload_regions(c, inputs[k::2]) for k, c in enumerate(ctxs)]
        results, lines = pipeline.run_hot_path_lanes(ctxs, batches)
        for c, b in zip(ctxs, batches):
            b.free(c)
    finally:
        for c in ctxs:
            c.close()
    key = lambda ls: sorted((c["chrom"], c["pos"], c["type"], c["svlen"], c["gt"]) for c in pipeline.parse_calls(ls))   # contigs are numbered per call
    assert key(lines) == key(one.lines) and len(lines) > 10
    assert sorted(l.split('\t')[4] for l in lines) == sorted(l.split('\t')[4] for l in one.lines)                     # ALT sequences


def test_bam_regions_straight_into_the_store(ctx, tmp_path):
    """N2: a haplotagged BAM -> crop + PS/HP grouping + 2-bit gather (pipeline.upload_bam_regions) gives the calls of the
    in-memory path fed with the same reads (BAM stores reverse-strand reads reverse-complemented; the assembler does not care)"""
    from tests import bam_writer as W
    rs = _regions([1, 4, 7])
    recs = []
    for r in rs:
        for h in (0, 1):
            for j, (pos, ops, rev) in enumerate(r.read_aln[h]):
                seq = r.reads[h][j]
                recs.append({"ref": 0, "pos": r.start + pos, "mapq": 60, "flag": 16 if rev else 0, "qname": "r%d_h%d_%d" % (r.index, h + 1, j),
                             "cigar": ops, "seq": (synth.revcomp(seq) if rev else seq).decode(),
                             "tags": [("PS", "I", r.start + 1), ("HP", "C", h + 1)]})
    recs.sort(key=lambda x: x["pos"])
    path = W.write_bam(str(tmp_path / "wgs.bam"), [("chr21", 46_000_000)], recs)
    batch = pipeline.upload_bam_regions(ctx, path, [(r.chrom, r.start + 1, r.start + len(r.ref)) for r in rs], [(r.start, r.ref) for r in rs])
    try:
        assert batch.set_kind == [1, 2] * 3 and batch.set_region == [0, 0, 1, 1, 2, 2]
        assert [int(x) for x in np.diff(batch.packed.set_start)] == [len(r.reads[h]) for r in rs for h in (0, 1)]
        res = pipeline.run_hot_path(ctx, batch)
    finally:
        batch.free(ctx)
    mem = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r) for r in rs])
    try:
        exp = pipeline.run_hot_path(ctx, mem)
    finally:
        mem.free(ctx)
    assert pipeline.parse_calls(res.lines) == pipeline.parse_calls(exp.lines) and len(res.lines) >= 6
    assert sorted(map(len, res.contig_batch)) == sorted(map(len, exp.contig_batch))


def test_run_stream_equals_sequential(ctx):
    """batches dealt to three lanes (whole batches in flight, a shared read store) give what one lane gives one after the other,
    in batch order, and on_result sees them in that order"""
    ra = [pipeline.region_from_synth(r) for r in _regions([2, 5, 9])]
    rb = [pipeline.region_from_synth(r) for r in _regions([3, 6])]
    ba, bb = pipeline.upload_regions(ctx, ra), pipeline.upload_regions(ctx, rb)
    lanes = [ctx, _lib.Context(0), _lib.Context(0)]
    try:
        seq = [pipeline.run_hot_path(ctx, b) for b in (ba, bb)]
        order = []
        got = pipeline.run_stream(lanes, [ba, bb, ba, ba, bb], on_result=lambda i, r: order.append(i))
        assert order == [0, 1, 2, 3, 4]
        for g, e in zip(got, [seq[0], seq[1], seq[0], seq[0], seq[1]]):
            assert g.lines == e.lines and g.raw_lines == e.raw_lines and list(g.contig_batch) == list(e.contig_batch)
        st = pipeline.run_stream(lanes[:2], [bb, ba, bb], static=True)
        assert [r.lines for r in st] == [seq[1].lines, seq[0].lines, seq[1].lines]
        assert pipeline.run_stream(lanes, []) == []
    finally:
        for c in lanes[1:]:
            c.close()
        ba.free(ctx); bb.free(ctx)


def test_bam_batches_through_the_lanes(ctx, tmp_path):
    """BAM + regions -> reader threads (bam_batches) -> lanes (run_stream uploads, runs and frees each HostBatch) -> the calls of the
    in-memory path, batch by batch in order"""
    from tests import bam_writer as W
    rs = _regions([1, 2, 4, 6, 7])
    recs = []
    for r in rs:
        for h in (0, 1):
            for j, (pos, ops, rev) in enumerate(r.read_aln[h]):
                seq = r.reads[h][j]
                recs.append({"ref": 0, "pos": r.start + pos, "mapq": 60, "flag": 16 if rev else 0, "qname": "r%d_h%d_%d" % (r.index, h + 1, j),
                             "cigar": ops, "seq": (synth.revcomp(seq) if rev else seq).decode(), "tags": [("PS", "I", r.start + 1), ("HP", "C", h + 1)]})
    recs.sort(key=lambda x: x["pos"])
    path = W.write_bam(str(tmp_path / "wgs.bam"), [("chr21", 46_000_000)], recs)
    regions = [(r.chrom, r.start + 1, r.start + len(r.ref)) for r in rs]
    wins = [(r.start, r.ref) for r in rs]
    lanes = [ctx, _lib.Context(0)]
    try:
        seen = []
        got = pipeline.run_stream(lanes, pipeline.bam_batches(path, regions, wins, batch=2, readers=2), on_result=lambda i, r: seen.append(i))
        assert seen == [0, 1, 2] and len(got) == 3
        for k, res in enumerate(got):
            part = rs[2 * k: 2 * k + 2]
            mem = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r) for r in part])
            try:
                exp = pipeline.run_hot_path(ctx, mem)
            finally:
                mem.free(ctx)
            assert pipeline.parse_calls(res.lines) == pipeline.parse_calls(exp.lines) and len(res.lines) >= 2 * len(part)

        def failing():
            yield pipeline.read_bam_regions(path, regions[:1], wins[:1])
            raise RuntimeError("reader failed")
        with pytest.raises(RuntimeError):
            pipeline.run_stream(lanes, failing())
    finally:
        lanes[1].close()


@pytest.mark.parametrize("size", [3000, 8000])
def test_tandem_duplication_end_to_end(ctx, size):
    """reads of a haplotype that carries a tandem duplication (a second copy of `size` bases: an INS) through assembly, contig
    alignment and the SV logic.  The 3 kb copy assembles into one contig; with 8 kb copies the string graph of 10-20 kb reads
    branches at the repeat (three unitigs, as hifiasm's would) and the contig that spans both copies carries the call -- which the
    aligner of round 2 would have refused (an 8 k x 16 k event).  One heterozygous INS of the exact length at the left-aligned
    position, nothing else"""
    import numpy as np
    rng = np.random.default_rng(99)
    width, at = 60000, 25000
    ref = np.frombuffer(synth.make_region(4000, width=width, depth_per_hap=0.1).ref, dtype=np.uint8).copy()
    hap1 = np.concatenate([ref[:at + size], ref[at:at + size], ref[at + size:]])
    r1 = synth._sample_reads(rng, hap1, 20.0, 10000, 20000, 0.002, 3000, synth._segments([(at, "INS", size)], width), [])
    r2 = synth._sample_reads(rng, ref, 20.0, 10000, 20000, 0.002, 3000, synth._segments([], width), [])
    b = pipeline.upload_regions(ctx, [pipeline.RegionInput("chr21", 0, ref.tobytes(), r1, r2, [], "dup")])
    try:
        res = pipeline.run_hot_path(ctx, b)
    finally:
        b.free(ctx)
    assert (res.set_status >= 0).all() and (res.contig_status >= 0).all()
    calls = [(c["type"], c["pos"], c["svlen"], c["gt"]) for c in pipeline.parse_calls(res.raw_lines)]
    assert len(calls) == 1 and calls[0][0] == "INS" and abs(calls[0][1] - at) <= 1 and calls[0][2] == size and calls[0][3] == "0/1", calls
