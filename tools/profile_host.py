"""cProfile of the host side of one bench step (needs a GPU): python tools/profile_host.py [n_regions]"""
import cProfile, pstats, sys, time, gc
sys.path.insert(0, ".")
from focalsv_amd import _lib, pipeline, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
regions = [synth.make_region(i, start=i * 60000) for i in range(n)]
ctx = _lib.Context(0)
batch = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r) for r in regions])
pipeline.run_hot_path(ctx, batch)
gc.collect(); gc.freeze()
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
res = pipeline.run_hot_path(ctx, batch)
pr.disable()
print("step", time.perf_counter() - t, "asm", res.asm_stats["ms_total"], "aln", res.aln_stats["ms_total"], res.host_ms)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
