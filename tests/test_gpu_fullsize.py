"""BASELINE.json configs[1] at full size -- 256 synthetic 50 kb regions, 512 read sets, one call -- checked through properties that do
not need the oracle on every set: the planted SVs come back (+-1 bp of the left-aligned truth, exact SVLEN, genotype), nothing else is
called, every read set gives one contig of its haplotype's length (+-3 bases at the ends), a second run reproduces contigs and VCF
text exactly (no ordering / race effects), two lanes over halves give the same calls, and a seeded sample of the sets is compared
with the oracle bit for bit."""
import random

import numpy as np
import pytest

from focalsv_amd import _lib, pipeline, synth
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
N = 256


@pytest.fixture(scope="module")
def full():
    regions = [synth.make_region(i, start=i * 60000) for i in range(N)]
    with _lib.Context(0) as ctx:
        batch = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r) for r in regions])
        try:
            a = pipeline.run_hot_path(ctx, batch)
            b = pipeline.run_hot_path(ctx, batch)
        finally:
            batch.free(ctx)
        yield regions, a, b, ctx


def test_planted_svs_and_nothing_else(full):
    regions, a, _, _ = full
    calls = pipeline.parse_calls(a.lines)
    truth = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for r in regions for t in r.truth]
    tp, fp, fn, gt_ok = pipeline.match_truth(calls, truth, bp_tol=1, len_tol=0.0, left_shift_ok=0)
    assert (tp, fp, fn, gt_ok) == (len(truth), 0, 0, len(truth)) and len(truth) >= 2 * N


def test_one_contig_per_read_set_of_haplotype_length(full):
    regions, a, _, _ = full
    assert not a.set_status.any() and not a.contig_status.any()
    seen = {}
    for i, (ri, hp) in enumerate(zip(a.contig_region, a.contig_hp)):
        seen.setdefault((ri, hp), []).append(a.contig_batch.length(i))
    assert len(seen) == 2 * N
    for (ri, hp), lens in seen.items():
        assert len(lens) == 1 and abs(lens[0] - len(regions[ri].haps[hp - 1])) <= 3, (ri, hp, lens)


def test_second_run_is_identical(full):
    _, a, b, _ = full
    assert a.lines == b.lines and a.raw_lines == b.raw_lines
    assert list(a.contig_batch) == list(b.contig_batch) and a.contig_region == b.contig_region


def test_sampled_sets_equal_the_oracle(full):
    regions, a, _, _ = full
    rng = random.Random(17)
    index = {(ri, hp): i for i, (ri, hp) in enumerate(zip(a.contig_region, a.contig_hp))}
    for ri in rng.sample(range(N), 3):
        hp = rng.choice([1, 2])
        contigs, _ = O.assemble(regions[ri].reads[hp - 1])
        assert len(contigs) == 1 and a.contig_batch[index[(ri, hp)]] == contigs[0], (ri, hp)


def test_lanes_over_halves_give_the_same_calls(full):
    regions, a, _, ctx = full
    inputs = [pipeline.region_from_synth(r) for r in regions]
    other = _lib.Context(0)
    halves = [pipeline.upload_regions(c, inputs[k::2]) for k, c in enumerate((ctx, other))]
    try:
        _, lines = pipeline.run_hot_path_lanes([ctx, other], halves)
    finally:
        for h, c in zip(halves, (ctx, other)):
            h.free(c)
        other.close()
    key = lambda c: (c["chrom"], c["pos"], c["type"], c["svlen"], c["gt"])
    assert sorted(map(key, pipeline.parse_calls(lines))) == sorted(map(key, pipeline.parse_calls(a.lines)))
