import cProfile, pstats, sys, time
sys.path.insert(0, ".")
from focalsv_amd import _lib, pipeline, synth
n = 256
regions = [synth.make_region(i, start=i * 60000) for i in range(n)]
ctx = _lib.Context(0)
batch = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r) for r in regions])
pipeline.run_hot_path(ctx, batch)
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
res = pipeline.run_hot_path(ctx, batch)
pr.disable()
print("step", time.perf_counter() - t, "asm", res.asm_stats["ms_total"], "aln", res.aln_stats["ms_total"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
