"""K5 on the GPU (through the C ABI) vs the reference-minted golden vectors and the CPU oracle."""
import json
import os
import random

import numpy as np
import pytest

from focalsv_amd import _lib
from tests import oracle_lib as O
from tests.kernel_cases import tasks_from_cases, usable

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    with _lib.Context(0) as c:
        yield c


def test_k5_golden(ctx, golden_dir):
    cases = [c for c in json.load(open(os.path.join(golden_dir, "bpm_k5.json")))["cases"] if usable(c)]
    assert len(cases) > 400
    words, tasks = tasks_from_cases(cases)
    res = ctx.bpm_windows(words, tasks)
    for c, r in zip(cases, res):
        assert int(r["err"]) == c["err"], c
        if c["err"] >= 0:
            assert int(r["end_site"]) == c["end_site"], c


def test_k5_random_vs_oracle(ctx):
    rng = random.Random(5)
    cases = []
    for i in range(20000):
        n = 375 if i % 3 else rng.randint(1, 375)
        k = 15 if n == 375 and i % 2 else min(31, max(1 if n >= 4 else 0, int(n * 0.04)) * rng.choice([1, 2]))
        core = "".join(rng.choice("ACGT") for _ in range(n + 2 * k + 40))
        x = core[20 + k: 20 + k + n]
        other = []
        rate = rng.choice([0.0, 0.003, 0.01, 0.04, 0.1])
        for ch in core:
            r = rng.random()
            if r < rate / 3:
                other.append(rng.choice("ACGT"))
            elif r < 2 * rate / 3:
                other.append(ch + rng.choice("ACGT"))
            elif r < rate:
                pass
            else:
                other.append(ch)
        other = "".join(other)
        d = rng.choice([0, 0, 1, -1, 2, -3])
        y = other[20 + d: 20 + d + n + 2 * k]
        if len(y) < n + 2 * k:
            y = y + "N" * (n + 2 * k - len(y))
        cases.append({"k": k, "x": x, "y": y})
    cases = [c for c in cases if usable(c)]
    words, tasks = tasks_from_cases(cases)
    res = ctx.bpm_windows(words, tasks)
    n_hit = 0
    for c, r in zip(cases, res):
        site, err = O.bpm(c["x"], c["y"], c["k"])
        assert int(r["err"]) == err, c
        if err >= 0:
            n_hit += 1
            assert int(r["end_site"]) == site, c
    assert n_hit > 5000


def test_k5_invalid_windows_are_reported_not_run(ctx):
    words, off, lens = _lib.pack_reads(["ACGT" * 100, "ACGT" * 100])
    tasks = np.zeros(3, dtype=_lib.WTASK_DTYPE)
    tasks[0] = (off[0], off[1], 0, -1, 400, 375, 15, 0, 0, 0)    # y_start < 0
    tasks[1] = (off[0], off[1], 0, 400, 400, 375, 15, 0, 0, 1)   # y_start past the read
    tasks[2] = (off[0], off[1], 0, 200, 400, 375, 15, 0, 0, 2)   # too little of y left
    res = ctx.bpm_windows(words, tasks)
    assert all(int(r["err"]) == -1 and int(r["end_site"]) == -1 for r in res)


def test_k5_reverse_strand(ctx):
    rng = random.Random(9)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    x = "".join(rng.choice("ACGT") for _ in range(375))
    y_fwd = "".join(rng.choice("ACGT") for _ in range(30)) + x[:100] + x[101:] + "".join(rng.choice("ACGT") for _ in range(30))
    y_read = "".join(comp[c] for c in reversed(y_fwd))  # stored read is the reverse complement
    words, off, lens = _lib.pack_reads([x, y_read])
    tasks = np.zeros(1, dtype=_lib.WTASK_DTYPE)
    tasks[0] = (off[0], off[1], 0, 30, len(y_read), 375, 15, 1, 0, 0)
    r = ctx.bpm_windows(words, tasks)[0]
    ypad = y_fwd[15: 15 + 405]
    assert (int(r["end_site"]), int(r["err"])) == O.bpm(x, ypad, 15)
    assert int(r["err"]) == 1


def test_k5_exact_diagonal_shortcut_in_repeats(ctx):
    """windows that match exactly on the predicted diagonal take a shortcut in the kernel; in a tandem repeat other end
    sites reach distance 0 too, and the reference's end-site rule must still come out (oracle = reference restatement)"""
    rng = random.Random(11)
    cases = []
    for i in range(400):
        unit = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 7)))
        n = rng.choice([375, 375, 200, 16, 17, 33])
        k = rng.choice([0, 1, 8, 15]) if n >= 33 else rng.choice([0, 1])
        core = (unit * (2 * (n + 2 * k) // len(unit) + 4))
        x = core[k: k + n]
        y = core[: n + 2 * k]
        if i % 5 == 0:   # one substitution somewhere: must not take the shortcut
            p = rng.randrange(n)
            x = x[:p] + "ACGT"[("ACGT".index(x[p]) + 1) % 4] + x[p + 1:]
        cases.append({"k": k, "x": x, "y": y})
    cases = [c for c in cases if usable(c)]
    words, tasks = tasks_from_cases(cases)
    res = ctx.bpm_windows(words, tasks)
    for c, r in zip(cases, res):
        site, err = O.bpm(c["x"], c["y"], c["k"])
        assert (int(r["err"]), int(r["end_site"]) if err >= 0 else -1) == (err, site if err >= 0 else -1), c


def test_k5_wide_bands_vs_oracle(ctx):
    """bands above 63 rows (k up to 95, BASELINE configs[4]: ONT-profile windows ~20 % apart): the wide-band kernel against the
    256-bit restatement (itself checked against a plain DP, tests/test_oracle_bpm.py); a list that holds one wide task runs
    entirely through the wide kernel, so narrow thresholds are compared there too"""
    from tests.test_oracle_bpm import _noisy_cases
    cases = [c for c in _noisy_cases(21, 6000, [15, 31, 40, 63, 80, 93, 95]) if usable(c)]
    cases[0]["k"] = min(cases[0]["k"], 31)
    words, tasks = tasks_from_cases(cases)
    res = ctx.bpm_windows(words, tasks)
    n_hit = n_wide = 0
    for c, r in zip(cases, res):
        site, err = O.bpm(c["x"], c["y"], c["k"])
        assert int(r["err"]) == err, c
        if err >= 0:
            n_hit += 1
            n_wide += c["k"] > 31
            assert int(r["end_site"]) == site, c
    assert n_hit > 2000 and n_wide > 1000
