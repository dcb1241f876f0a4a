"""BASELINE.json configs[2] at full size: every one of the 377 chr21 lines of the reference's auto-mode BED
(test/SV_Regions_HG002_HIFI_L1_FocalSV-auto.bed, widths 14 000 .. 467 280 bp; tests/golden/bed_chr21_regions.json), synthetic
30x HiFi-like reads laid over each region, through the whole hot path in one call -- plus the widest line of the whole-genome BED
(1 146 440 bp, configs[3]).  Checks are properties (planted truth, one contig per read set, nothing else called) and, on a stated
sample, equality with the CPU path (oracle contigs + oracle alignments through the same host logic)."""
import json
import os
import time

import numpy as np
import pytest

from focalsv_amd import _lib, pipeline, synth
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu

MARGIN = 15000      # `samtools view bam chr:s-e` (1_crop_bam.py:74) keeps whole reads that touch the region: a read set spans about
                    # a read length beyond either end of its BED line


@pytest.fixture(scope="module")
def chr21(golden_dir):
    bed = json.load(open(os.path.join(golden_dir, "bed_chr21_regions.json")))["chr21"]
    assert len(bed) == 377 and max(b - a for a, b in bed) == 467280
    t0 = time.time()
    regions = [synth.make_region(3000 + i, width=b - a + 2 * MARGIN, start=max(0, a - MARGIN)) for i, (a, b) in enumerate(bed)]
    t_synth = time.time() - t0
    with _lib.Context(0) as ctx:
        t0 = time.time()
        batch = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r) for r in regions])
        t_up = time.time() - t0
        try:
            pipeline.run_hot_path(ctx, batch)            # warm-up: workspace allocation
            t0 = time.time()
            res = pipeline.run_hot_path(ctx, batch)
            t_run = time.time() - t0
            # per-width timing: regions grouped by width, each group through the hot path on its own
            order = sorted(range(len(regions)), key=lambda i: len(regions[i].ref))
            groups = {"narrowest 64 (14.0-14.1 kb lines)": order[:64], "median 64": order[156:220], "widest 16": order[-16:], "the 467 kb line": order[-1:]}
            timing = {}
            for name, idx in groups.items():
                gb = pipeline.upload_regions(ctx, [pipeline.region_from_synth(regions[i]) for i in idx])
                try:
                    pipeline.run_hot_path(ctx, gb)
                    t1 = time.time()
                    gr = pipeline.run_hot_path(ctx, gb)
                    dt = time.time() - t1
                finally:
                    gb.free(ctx)
                timing[name] = {"regions": len(idx), "bed_width_min": min(bed[i][1] - bed[i][0] for i in idx), "bed_width_max": max(bed[i][1] - bed[i][0] for i in idx),
                                "read_bases": int(sum(r.work for r in gb.regions)), "seconds": round(dt, 4), "regions_per_s": round(len(idx) / dt, 1),
                                "assembly_ms": round(gr.asm_stats["ms_total"], 2), "align_ms": round(gr.aln_stats.get("ms_total", 0.0), 2)}
        finally:
            batch.free(ctx)
    doc = {"what": "BASELINE.json configs[2]: all 377 chr21 lines of the reference's auto-mode BED, synthetic 30x HiFi-like reads (seed 4000+i), one MI355X, one lane",
           "regions": len(regions), "read_bases": int(sum(sum(map(len, r.reads[0])) + sum(map(len, r.reads[1])) for r in regions)),
           "seconds_whole_bed_one_call": round(t_run, 3), "regions_per_s": round(len(regions) / t_run, 1),
           "assembly_ms": round(res.asm_stats["ms_total"], 1), "align_ms": round(res.aln_stats.get("ms_total", 0.0), 1), "host_ms": res.host_ms,
           "synth_seconds": round(t_synth, 1), "upload_seconds": round(t_up, 2), "groups": timing}
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(doc, open(os.path.join("gpurun_out", "chr21_bed_timing.json"), "w"), indent=1)
    return regions, res


def test_every_planted_sv_of_the_377_regions(chr21):
    regions, res = chr21
    assert (res.contig_status == 0).all() and not res.failed_regions
    assert all(int(s) in (0, _lib.W_NO_LAYOUT) for s in res.set_status)
    calls = pipeline.parse_calls(res.lines)
    truth = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for r in regions for t in r.truth]
    tols = [synth.position_tolerance(r, t) for r in regions for t in r.truth]
    tp, fp, fn, gt_ok = pipeline.match_truth(calls, truth, bp_tol=1, len_tol=0.0, tols=tols)
    # a 14 kb line + margins is a 44 kb window; at 15x per haplotype a handful of read sets lay out as a chain of fewer than four
    # reads at one end, which hifiasm cuts as a tip (FSV_W_NO_LAYOUT): an SV there is not callable from that haplotype
    no_layout = {pipeline_set_region(res, s) for s, st in enumerate(res.set_status) if int(st) & _lib.W_NO_LAYOUT}
    lost = sum(1 for r_i, r in enumerate(regions) if r_i in no_layout for t in r.truth)
    assert fp == 0 and fn <= lost and tp >= len(truth) - lost, (tp, fp, fn, len(truth), lost)
    assert gt_ok >= tp - 2 * lost
    assert len(truth) > 800


def pipeline_set_region(res, s):
    return s // 2          # upload_regions files two read sets per region (hp1, hp2) when there are no unphased reads


def test_one_contig_of_haplotype_length_per_read_set(chr21):
    regions, res = chr21
    per = {}
    for ri, hp, c in res.contigs:
        per.setdefault((ri, hp), []).append(len(c))
    whole = 0
    for ri, r in enumerate(regions):
        for h in (0, 1):
            got = per.get((ri, h + 1), [])
            if len(got) == 1 and abs(got[0] - len(r.haps[h])) <= 2:
                whole += 1
    assert whole >= 2 * len(regions) - 8, whole           # all but a few low-coverage ends


def test_sample_equals_the_cpu_path(chr21):
    """SV calls of the GPU path against the CPU path (BASELINE: "SV F1 vs CPU pipeline") on the five narrowest, the median and the
    75th-percentile region: the raw VCF body is the same text"""
    from focalsv_amd.dippav import signatures as S
    from focalsv_amd.dippav.variant_call import WindowedRef, call_chromosome
    regions, res = chr21
    order = sorted(range(len(regions)), key=lambda i: len(regions[i].ref))
    sample = sorted(order[:5] + [order[len(order) // 2], order[len(order) * 3 // 4]])
    recs, contig_seq, cnt = [], {}, {1: 0, 2: 0}
    for i in sample:
        r = regions[i]
        for h in (0, 1):
            for c in O.assemble(r.reads[h])[0]:
                name = "contig_hp%d_%d" % (h + 1, cnt[h + 1]); cnt[h + 1] += 1
                a = O.align_contig(c, r.ref)
                contig_seq[name] = c.decode()
                if a:
                    recs.append(S.AlignedSegment(r.chrom, r.start + a["ref_start"], r.start + a["ref_end"], a["cigar"], name, bool(a["rev"]), a["mapq"], None))
    recs.sort(key=lambda x: x.pos)
    ref = WindowedRef()
    for i in sample:
        ref.add(regions[i].start, regions[i].ref.decode())
    _, body = call_chromosome(recs, "chr21", ref, contig_seq, 'CCS')
    # the same regions through the GPU path as a batch of their own (neighbouring BED lines overlap once the margins are added, so
    # calls cannot be told apart by position in the 377-region run)
    with _lib.Context(0) as ctx:
        b = pipeline.upload_regions(ctx, [pipeline.region_from_synth(regions[i]) for i in sample])
        try:
            got = pipeline.run_hot_path(ctx, b)
        finally:
            b.free(ctx)
    assert got.raw_lines == body and len(body) >= 10
    # and inside the full run: the same calls (contig numbering differs)
    key = lambda ls: sorted((c["chrom"], c["pos"], c["type"], c["svlen"], c["gt"]) for c in pipeline.parse_calls(ls))
    assert set(key(body)) <= set(key(res.raw_lines))


def test_the_widest_whole_genome_line():
    """the 1 146 440 bp line of the whole-genome BED (configs[3]): 2 400 reads in two sets, 8.5 M read pairs, a 1.18 Mb reference
    window -- beyond round 1's aligner limit (~760 kb) and far above the read counts it was tried on.  One contig per haplotype,
    both planted SVs, the aligner's CIGARs equal to the oracle's."""
    r = synth.make_region(9001, width=1146440 + 2 * MARGIN, start=0)
    with _lib.Context(0) as ctx:
        b = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r)])
        try:
            res = pipeline.run_hot_path(ctx, b)
        finally:
            b.free(ctx)
        assert list(res.set_status) == [0, 0] and list(res.contig_status) == [0, 0]
        assert sorted(len(c) for _, _, c in res.contigs) == sorted(len(h) for h in r.haps)
        calls = pipeline.parse_calls(res.lines)
        truth = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for t in r.truth]
        assert pipeline.match_truth(calls, truth, 1, 0.0)[:3] == (len(truth), 0, 0)
        rec, cigar, status = ctx.align_batch([r.haps[0], r.haps[1]], [0, 0], [r.ref])
        assert list(status) == [0, 0]
        for rr in rec:
            i = int(rr["contig"])
            o = O.align_contig(r.haps[i], r.ref)
            assert list(cigar[int(rr["cigar_off"]): int(rr["cigar_off"]) + int(rr["n_cigar"])]) == list(o["raw"])


def test_a_batch_beyond_the_budget_is_split_by_sets(monkeypatch):
    """fsv_assemble_batch cuts a batch that exceeds its workspace budget (or the 2^31 pair / task indices) into runs of consecutive
    read sets: same contigs, same calls as the one-pass run"""
    rs = [synth.make_region(40 + i, start=i * 60000) for i in range(12)]
    inputs = [pipeline.region_from_synth(r) for r in rs]
    with _lib.Context(0) as ctx:
        b = pipeline.upload_regions(ctx, inputs)
        try:
            one = pipeline.run_hot_path(ctx, b)
            monkeypatch.setenv("FSV_ASM_BUDGET_GB", "0.6")          # a few regions per pass
            many = pipeline.run_hot_path(ctx, b)
        finally:
            b.free(ctx)
    assert [c for c in one.contigs] == [c for c in many.contigs]
    assert one.raw_lines == many.raw_lines and one.lines == many.lines and len(one.lines) > 20
    assert list(one.set_status) == list(many.set_status)
    assert many.asm_stats["n_windows"] == one.asm_stats["n_windows"]


def test_whole_genome_bed_sample_through_the_region_queue():
    """BASELINE.json configs[3] on a sample: every 13th line of the whole-genome auto-mode BED (2 000 of 26 834, the 1 146 440 bp
    line included) through `bench.py --workload bed` -- RegionQueue, three lanes, read stores uploaded per batch, one VCF gather.
    Every planted SV back at +-1 bp with exact length and genotype, nothing else called, no region failed.  (The run over all
    26 834 lines is committed under profiles/; it needs minutes of host-side read synthesis.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "bed", "--bed", "genome", "--bed-limit", "2000"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    doc = json.loads(p.stdout.strip().splitlines()[-1])
    assert doc["n_gpus"] == 1 and doc["ranks"][0]["regions"] == 2000 and doc["regions_failed"] == 0
    assert "1176440" in doc["config"]["workload"]          # the widest line + its margins is in the sample
    sv = doc["sv_vs_truth"]
    assert sv["truth"] >= 4000 and sv["tp"] == sv["truth"] == sv["gt_ok"] and sv["fp"] == 0 and sv["fn"] == 0, sv
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    json.dump(doc, open(os.path.join(root, "gpurun_out", "bed_genome2000_n1.json"), "w"), indent=1)


def _nccl_gather_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    lines = {0: ["chr21\t500\ta\tA\tAT\t20\tPASS\tSVLEN=1;SVTYPE=INS\tGT\t0/1\n", "chr2\t9\tb\tAT\tA\t20\tPASS\tSVLEN=-1;SVTYPE=DEL\tGT\t1/1\n"],
             1: ["chr21\t20\tc\tA\tAT\t20\tPASS\tSVLEN=1;SVTYPE=INS\tGT\t0/1\n"]}.get(rank, [])
    out = pipeline.gather_vcf(lines, device=torch.device("cuda", rank))
    empty = pipeline.gather_vcf([] if rank else ["chrX\t1\td\tA\tAT\t20\tPASS\tSVLEN=1;SVTYPE=INS\tGT\t0/1\n"], device=torch.device("cuda", rank))
    q.put((rank, out, empty))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_vcf_over_rccl_with_two_gpus():
    """the one exchange step of the multi-GPU path on its real backend: two ranks, one GPU each, `nccl` (= RCCL over xGMI).
    Needs two visible GPUs: skipped on the one-GPU box (the gloo rehearsal of the same function is tests/test_distributed.py)."""
    import socket
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL between two ranks); this box has %d" % torch.cuda.device_count())
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_nccl_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = {r: (o, e) for r, o, e in (q.get(timeout=300) for _ in range(2))}
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    assert res[0] == res[1]
    assert [l.split('\t')[2] for l in res[0][0]] == ["b", "c", "a"]  # chr2 before chr21, then by position
    assert [l.split('\t')[2] for l in res[0][1]] == ["d"]
