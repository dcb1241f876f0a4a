"""exploration: CLR-profile reads (synth profile "clr": 12 % error, insertion-rich) through the hot path with the ONT parameter set"""
import sys, os, time
sys.path.insert(0, os.getcwd())
from focalsv_amd import _lib, pipeline, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
regions = [synth.make_region(i, width=50000, profile="clr", start=i * 60000) for i in range(n)]
with _lib.Context(0) as ctx:
    b = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r) for r in regions])
    t = time.time()
    res = pipeline.run_hot_path(ctx, b, asm_params=ctx.ont_asm_params(), data_type='CLR')
    print("seconds", time.time() - t)
    b.free(ctx)
print("set_status", list(res.set_status), "contig_status", list(res.contig_status))
per = {}
for ri, hp, c in res.contigs:
    per.setdefault((ri, hp), []).append(len(c))
print({k: v for k, v in sorted(per.items())})
print([len(h) for r in regions for h in r.haps])
calls = pipeline.parse_calls(res.lines)
truth = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for r in regions for t in r.truth]
print("truth", len(truth), "loose", pipeline.match_truth(calls, truth, bp_tol=20, len_tol=0.02, left_shift_ok=2000), "strict", pipeline.match_truth(calls, truth, bp_tol=1, len_tol=0.0))
