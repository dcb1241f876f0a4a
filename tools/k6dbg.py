import json, os, sys
sys.path.insert(0, '/root/repo')
from focalsv_amd import _lib
from tests import oracle_lib as O
from tests.kernel_cases import strip_pad, tasks_from_cases, usable
cases = [c for c in json.load(open('/root/repo/tests/golden/bpm_k6.json'))["cases"] if usable(c)]
with _lib.Context(0) as ctx:
    words, tasks = tasks_from_cases(cases)
    res, paths = ctx.bpm_paths(words, tasks)
bad = 0
for i, (c, r, p) in enumerate(zip(cases, res, paths)):
    site, err, start, path = O.bpm_path(c["x"], c["y"], c["k"])
    if err < 0: continue
    raw = bytes(path[::-1])
    ops = _lib.path_ops(p)
    padl = strip_pad(c["y"])[0]
    if ops != raw or int(p["ry_start"]) != start - padl:
        bad += 1
        if bad <= 5:
            print(i, "k", c["k"], "n", len(c["x"]), "err", err, "padl", padl, "start", start - padl, int(p["ry_start"]))
            print(" raw", "".join(map(str, raw[:70])))
            print(" gpu", "".join(map(str, ops[:70])))
print("raw-path mismatches", bad, "of", len(cases))
