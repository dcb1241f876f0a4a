"""K6 on the GPU (fsv_bpm_paths through the C ABI): alignment path + generate_cigar against the reference-minted golden
vectors (tests/golden/bpm_k6.json) and the CPU oracle on random windows, single-indel windows in repeats included."""
import json
import os
import random

import pytest

from focalsv_amd import _lib
from tests import oracle_lib as O
from tests.kernel_cases import strip_pad, tasks_from_cases, usable

pytestmark = pytest.mark.gpu
OPS = {"M": 0, "X": 1, "I": 2, "D": 3}


@pytest.fixture(scope="module")
def ctx():
    with _lib.Context(0) as c:
        yield c


def expected(c):
    """oracle, in the order the assembler uses (Levenshtein_distance.h:516-531): K5's result first; distance 0 or a gap-free
    placement with exactly that many mismatches (try_cigar) needs no DP; else the banded DP with traceback.
    -> (err after generate_cigar, start, end, ops start-to-end) or None when no alignment within k"""
    site, err = O.bpm(c["x"], c["y"], c["k"])
    if err < 0:
        return None
    fast = (site - len(c["x"]) + 1, bytes(len(c["x"]))) if err == 0 else O.try_cigar(c["x"], c["y"], site, err)
    if fast is not None:
        start, path = fast
    else:
        site2, err2, start, path = O.bpm_path(c["x"], c["y"], c["k"], wide=c["k"] > 31)
        assert (site2, err2) == (site, err)
    st, en, er, cg = O.generate_cigar(path, c["x"], c["y"], start, site, err)
    ops, num = [], ""
    for ch in cg:
        if ch.isdigit():
            num += ch
        else:
            ops += [OPS[ch]] * int(num); num = ""
    if len(ops) > 416:
        return "too long"        # more ops than a path record holds: the window is left without a path (wide bands only)
    return er, st, en, bytes(ops)


def check(ctx, cases):
    cases = [c for c in cases if usable(c)]
    words, tasks = tasks_from_cases(cases)
    res, paths = ctx.bpm_paths(words, tasks)
    n = 0
    for c, r, p in zip(cases, res, paths):
        e = expected(c)
        if e is None:
            assert int(r["err"]) < 0 and int(p["state"]) == 0, c
            continue
        if e == "too long":
            assert int(p["state"]) == 0, c
            continue
        n += 1
        padl = strip_pad(c["y"])[0]
        er, st, en, ops = e
        assert int(p["state"]) == 1, c
        assert (int(p["err"]), int(p["ry_start"]), int(p["ry_end"])) == (er, st - padl, en - padl), (c, e[:3])
        assert _lib.path_ops(p) == ops, c
    return n


def test_k6_golden(ctx, golden_dir):
    """the goldens were minted from the reference's pure DP (old_error = -1); wherever the assembler's gap-free shortcut does not
    apply, or gives the same path, the GPU must reproduce the reference's own output"""
    cases = [c for c in json.load(open(os.path.join(golden_dir, "bpm_k6.json")))["cases"] if usable(c)]
    assert check(ctx, cases) > 150
    words, tasks = tasks_from_cases(cases)
    res, paths = ctx.bpm_paths(words, tasks)
    n_ref = 0
    for c, p in zip(cases, paths):
        if c["err"] < 0:
            continue
        site, err = O.bpm(c["x"], c["y"], c["k"])
        if err > 0 and O.try_cigar(c["x"], c["y"], site, err) is not None:
            continue   # the shortcut may pick another of the equally good paths
        ops, num = [], ""
        for ch in c["cigar"]:
            if ch.isdigit():
                num += ch
            else:
                ops += [OPS[ch]] * int(num); num = ""
        assert _lib.path_ops(p) == bytes(ops) and int(p["err"]) == c["cigar_err"], c
        n_ref += 1
    assert n_ref > 100


def test_k6_random_vs_oracle(ctx):
    rng = random.Random(6)
    cases = []
    for i in range(6000):
        n = 375 if i % 3 else rng.randint(8, 375)
        k = 15 if n == 375 else max(1, int(n * 0.04))
        core = "".join(rng.choice("ACGT") for _ in range(n + 2 * k + 40))
        x = core[20 + k: 20 + k + n]
        rate = rng.choice([0.0, 0.003, 0.003, 0.01, 0.03])
        other = []
        for ch in core:
            r = rng.random()
            if r < rate / 3: other.append(rng.choice("ACGT"))
            elif r < 2 * rate / 3: other.append(ch + rng.choice("ACGT"))
            elif r < rate: pass
            else: other.append(ch)
        other = "".join(other)
        d = rng.choice([0, 0, 1, -1, 2])
        y = other[20 + d: 20 + d + n + 2 * k]
        if len(y) < n + 2 * k:
            y = y + "N" * (n + 2 * k - len(y))
        cases.append({"k": k, "x": x, "y": y})
    assert check(ctx, cases) > 3000


def test_k6_single_indels_in_repeats(ctx):
    """one inserted or deleted base inside low-complexity sequence: many equally good gap placements, the reference's walk-back
    and greedy left shift decide"""
    rng = random.Random(66)
    cases = []
    for i in range(3000):
        n = rng.choice([375, 375, 120, 40])
        k = 15 if n == 375 else max(1, int(n * 0.04))
        unit = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 4)))
        core = []
        while len(core) < n + 2 * k + 8:
            if rng.random() < 0.5:
                core += list(unit * rng.randint(2, 12))
            else:
                core += [rng.choice("ACGT") for _ in range(rng.randint(1, 30))]
        core = "".join(core)[: n + 2 * k + 8]
        x = core[k: k + n]
        y = list(core)
        p = rng.randrange(k, k + n)
        if i % 2:
            del y[p]
        else:
            y.insert(p, rng.choice("ACGT"))
        y = "".join(y)[: n + 2 * k]
        if len(y) < n + 2 * k:
            y += "N" * (n + 2 * k - len(y))
        cases.append({"k": k, "x": x, "y": y})
    assert check(ctx, cases) > 2000


def test_k6_wide_bands_vs_oracle(ctx):
    """paths of wide-band windows (k up to 95): walk, generate_cigar and record identical to the oracle's; a path of more than
    416 ops leaves the window without a record"""
    from tests.test_oracle_bpm import _noisy_cases
    cases = _noisy_cases(22, 2500, [40, 63, 80, 93, 95], rates=(0.02, 0.1, 0.2, 0.25))
    assert check(ctx, cases) > 800
