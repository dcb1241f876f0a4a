import sys, time
sys.path.insert(0, ".")
from focalsv_amd import _lib, pipeline, synth
w = int(sys.argv[1]) if len(sys.argv) > 1 else 467280
t = time.time(); r = synth.make_region(900, width=w, start=1000000); print("gen", time.time() - t, len(r.reads[0]), len(r.reads[1]), flush=True)
ctx = _lib.Context(0)
batch = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r)])
for it in range(2):
    t = time.time(); res = pipeline.run_hot_path(ctx, batch); dt = time.time() - t
    print("step", round(dt, 3), "status", res.set_status, res.contig_status, "contigs", [(hp, len(c)) for _, hp, c in res.contigs], "haps", len(r.haps[0]), len(r.haps[1]), flush=True)
calls = pipeline.parse_calls(res.lines)
truth = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for t in r.truth]
print(pipeline.match_truth(calls, truth, 1, 0.02), [(c["type"], c["pos"] - r.start, c["svlen"], c["gt"]) for c in calls], [(t[1], t[2] - r.start, t[3], t[4]) for t in truth])
print({k: round(v, 1) for k, v in res.asm_stats.items() if k.startswith("ms_")}, {k: round(v, 1) for k, v in res.aln_stats.items() if k.startswith("ms_")})
