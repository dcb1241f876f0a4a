import sys
sys.path.insert(0, ".")
from focalsv_amd import _lib, pipeline, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
regions = [synth.make_region(i, start=i * 60000) for i in range(n)]
ctx = _lib.Context(0)
batch = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r) for r in regions])
res = pipeline.run_hot_path(ctx, batch)
calls = pipeline.parse_calls(res.lines)
raw = pipeline.parse_calls(res.raw_lines)
for r in regions:
    truth = [(r.chrom, t.svtype, r.start + t.pos, t.length, t.gt) for t in r.truth]
    mine = [c for c in calls if r.start <= c["pos"] < r.start + 60000]
    tp, fp, fn, gt = pipeline.match_truth(mine, truth, 1, 0.02, 2000)
    if fn or fp:
        rawm = [c for c in raw if r.start <= c["pos"] < r.start + 60000]
        nct = [(ri, hp, len(c)) for ri, hp, c in res.contigs if ri == r.index]
        print("region", r.index, "truth", [(t[1], t[2] - r.start, t[3], t[4]) for t in truth], "calls", [(c["type"], c["pos"] - r.start, c["svlen"], c["gt"]) for c in mine],
              "raw", [(c["type"], c["pos"] - r.start, c["svlen"], c["gt"]) for c in rawm], "contigs", nct, "hap lens", len(r.haps[0]), len(r.haps[1]))
