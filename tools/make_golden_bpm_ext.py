#!/usr/bin/env python3
"""Golden vectors for the extension alignment from the reference itself: alignment_extension / Reserve_Banded_BPM_Extension
(hifiasm-0.14 Levenshtein_distance.h:63-266), both directions, through oracle/_ref/ha14_kernels (`make -f oracle/ref.mk`)
-> tests/golden/bpm_ext.json.  The cases are what non_trim_error_rate (Correct.cpp:725-845) gives it: a window of up to 375 bases
against the other read's stretch with the doubled threshold on each side, where some way into the window an insertion, a deletion or
unrelated sequence begins -- and windows that align to the end, clean or noisy, with N padding at either end."""
import json, os, random, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ha14_kernels")


def rnd(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def noisy(rng, s, rate):
    out = []
    for ch in s:
        r = rng.random()
        if r < rate / 3:
            out.append(rng.choice([b for b in "ACGT" if b != ch]))
        elif r < 2 * rate / 3:
            out.append(ch); out.append(rng.choice("ACGT"))
        elif r < rate:
            pass
        else:
            out.append(ch)
    return "".join(out)


def main():
    rng = random.Random(20261005)
    cases = []
    for i in range(400):
        n = 375 if i % 3 else rng.randint(5, 374)
        k = rng.choice([31, 31, 30, 16, 8, 3]) if n >= 100 else max(1, min(31, n // 6))
        core = rnd(rng, n)
        kind = i % 5
        cut = rng.randint(0, n)
        if kind == 0:
            other = core                                                    # aligns to the end
        elif kind == 1:
            other = core[:cut] + rnd(rng, rng.choice([40, 100, 163, 400])) + core[cut:]     # y has an insertion
        elif kind == 2:
            other = core[:cut] + core[min(n, cut + rng.choice([40, 100, 200])):]            # y lacks a stretch
        elif kind == 3:
            other = core[:cut] + rnd(rng, n)                                # unrelated from cut on
        else:
            other = rnd(rng, cut) + core[cut:]                              # unrelated up to cut (the right-to-left case)
        other = noisy(rng, other, rng.choice([0.0, 0.0, 0.005, 0.02, 0.06]))
        drift = rng.choice([0, 0, 0, 1, -1, 4, -5])
        y = rnd(rng, k + 8)[: k + drift if k + drift > 0 else 0] + other + rnd(rng, n + 2 * k)
        y = y[: n + 2 * k]
        padl, padr = rng.choice([0, 0, 0, rng.randint(1, k)]), rng.choice([0, 0, 0, rng.randint(1, k)])
        y = "N" * padl + y[padl:]
        y = y[: len(y) - padr] + "N" * padr
        cases.append({"k": k, "dir": i % 2 if kind != 4 else 1, "x": core, "y": y})
    p = subprocess.run([HARNESS], input="\n".join(f"ext {c['k']} {c['dir']} {c['x']} {c['y']}" for c in cases) + "\n", capture_output=True, text=True, check=True)
    out = p.stdout.strip("\n").split("\n")
    assert len(out) == len(cases)
    for c, r in zip(cases, out):
        c["aligned"], c["err"], c["p_end"], c["t_end"] = map(int, r.split())
    json.dump({"source": "tools/make_golden_bpm_ext.py: alignment_extension of the reference's hifiasm-0.14 (oracle/_ref/ha14_kernels); aligned = bases of x covered "
                         "(0: none), err = distance there (-1: none); p_end / t_end as the reference returns them", "cases": cases},
              open(os.path.join(ROOT, "tests", "golden", "bpm_ext.json"), "w"), indent=0)
    print(len(cases), "cases;", sum(c["aligned"] == 0 for c in cases), "unaligned,", sum(c["aligned"] == len(c["x"]) for c in cases), "to the end")


main()
