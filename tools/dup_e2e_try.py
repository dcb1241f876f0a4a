"""exploration: a region whose haplotype 1 carries a tandem duplication (INS of a second copy) through the whole hot path"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from focalsv_amd import _lib, pipeline, synth
from focalsv_amd.dippav import signatures as S
size = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
rng = np.random.default_rng(99)
width = 60000
base = synth.make_region(4000, width=width, depth_per_hap=0.1)
ref = np.frombuffer(base.ref, dtype=np.uint8).copy()
at = 25000
hap1 = np.concatenate([ref[:at + size], ref[at:at + size], ref[at + size:]])   # the copy of [at, at+size) again
hap2 = ref.copy()
a1, a2 = [], []
seg1 = synth._segments([(at, "INS", size)], width)
seg2 = synth._segments([], width)
r1 = synth._sample_reads(rng, hap1, 20.0, 10000, 20000, 0.002, 3000, seg1, a1)
r2 = synth._sample_reads(rng, hap2, 20.0, 10000, 20000, 0.002, 3000, seg2, a2)
recs = []
inp = pipeline.RegionInput("chr21", 0, ref.tobytes(), r1, r2, [], "dup")
with _lib.Context(0) as ctx:
    b = pipeline.upload_regions(ctx, [inp])
    res = pipeline.run_hot_path(ctx, b)
    b.free(ctx)
print("set_status", list(res.set_status), "contig_status", list(res.contig_status))
print("contigs", [(hp, len(c)) for ri, hp, c in res.contigs], "hap lens", len(hap1), len(hap2))
print("calls", [(c["type"], c["pos"], c["svlen"], c["gt"]) for c in pipeline.parse_calls(res.raw_lines)])
