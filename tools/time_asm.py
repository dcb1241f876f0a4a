import sys, time
import numpy as np
sys.path.insert(0, ".")
from focalsv_amd import _lib, synth
from focalsv_amd.readsets import pack_sets
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
t = time.time()
regions = [synth.make_region(i) for i in range(n)]
print("gen", time.time() - t, flush=True)
sets = [rd for r in regions for rd in r.reads]
t = time.time(); b = pack_sets(sets); print("pack", time.time() - t, "reads", b.n_reads, "words", b.words.size, flush=True)
with _lib.Context(0) as ctx:
    print(ctx.device_info())
    d = ctx.upload(b.words)
    for it in range(3):
        t = time.time()
        contigs, cset, cnr, status = ctx.assemble_batch(d, b.word_off, b.read_len, b.set_start)
        dt = time.time() - t
        st = ctx.asm_stats()
        print(f"iter {it}: {dt:.3f}s  {n/dt:.1f} regions/s  contigs={len(contigs)}", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in st.items()}, flush=True)
    ok = 0
    for ri, r in enumerate(regions):
        for h in (0, 1):
            mine = [c for c, cs in zip(contigs, cset) if cs == 2 * ri + h]
            ok += (len(mine) == 1 and (mine[0] == r.haps[h] or synth.revcomp(mine[0]) == r.haps[h]))
    print("sets assembled exactly:", ok, "/", 2 * n, "status nonzero:", int((status != 0).sum()))
