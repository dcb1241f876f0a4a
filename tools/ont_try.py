#!/usr/bin/env python3
"""Exploration aid: ONT-profile regions (synth profile 'ont': 10 % error, reads U(10k, 30k)) through the whole hot path with
fsv_asm_ont_params, planted truth scored; optionally (--oracle N) the first N read sets against the CPU oracle."""
import sys, time
sys.path.insert(0, '.')
from focalsv_amd import _lib, synth, pipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
width = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
n_or = int(sys.argv[3]) if len(sys.argv) > 3 else 0
regs = [synth.make_region(i, width=width, profile='ont', start=i * 60000) for i in range(n)]
with _lib.Context(0) as ctx:
    p = ctx.ont_asm_params()
    b = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r) for r in regs])
    res = pipeline.run_hot_path(ctx, b, asm_params=p)
    t0 = time.time()
    res = pipeline.run_hot_path(ctx, b, asm_params=p)
    dt = time.time() - t0
    b.free(ctx)
print("hot path %.2fs for %d regions (%.1f regions/s); set status %s" % (dt, n, n / dt, sorted(set(int(s) for s in res.set_status))))
print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.asm_stats.items() if not isinstance(v, dict)})
print({k: round(v["ms"], 1) for k, v in res.asm_stats["kernels"].items()})
per = {}
for ri, hp, c in res.contigs:
    per.setdefault((ri, hp), []).append(len(c))
print("contigs per set:", sorted(set(len(v) for v in per.values())), "missing sets:", [(ri, hp) for ri in range(n) for hp in (1, 2) if (ri, hp) not in per])
calls = pipeline.parse_calls(res.lines)
truth = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for r in regs for t in r.truth]
print("strict  (1 bp, exact len):", pipeline.match_truth(calls, truth, 1, 0.0))
print("relaxed (20 bp or left shift in repeats, 2 % len):", pipeline.match_truth(calls, truth, 20, 0.02, left_shift_ok=2000), "truth", len(truth), "calls", len(calls))
if n_or:
    from tests import oracle_lib as O
    po = O.default_params()
    po.k, po.w, po.hpc, po.bw_ec, po.bw_final, po.win_rate_pm, po.k_cap, po.accept_err_pm, po.bw_rechain, po.min_contig_reads = 15, 15, 0, 150, 50, 250, 95, 300, 50, 2
    for si in range(n_or):
        r = regs[si // 2]
        oc, _ = O.assemble(r.reads[si % 2], po)
        mine = [c for ri, hp, c in res.contigs if ri == si // 2 and hp == si % 2 + 1]
        print("set", si, "contigs equal the oracle's:", mine == oc, [len(c) for c in oc], flush=True)
