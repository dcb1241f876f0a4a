#!/usr/bin/env python3
"""Mint kernel-level golden vectors from the *reference itself*.

Runs only in the build container (needs oracle/_ref/ha14_kernels, built by
`make -f oracle/ref.mk` from /root/reference/software/hifiasm-0.14).  Writes
inputs + the reference's answers to tests/golden/*.json; only that data is
committed.  Vectors:
  bpm_k5.json     Reserve_Banded_BPM        (Levenshtein_distance.h:274-461)
  bpm_k6.json     Reserve_Banded_BPM_PATH + generate_cigar (…:511-888; Correct.cpp:1387-1536)
  sketch.json     ha_sketch w=51 k=51 HPC   (sketch.cpp:39-137)
  ksw_extz2.json  ksw_extz2_sse             (ksw2_extz2_sse.c:23-305)
"""
import json
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ha14_kernels")
OUT = os.path.join(ROOT, "tests", "golden")


def ask(lines):
    p = subprocess.run([HARNESS], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True)
    out = p.stdout.strip("\n").split("\n")
    assert len(out) == len(lines), (len(out), len(lines))
    return out


def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def mutate(rng, s, rate, hp_bias=False):
    out = []
    i = 0
    while i < len(s):
        r = rng.random()
        if r < rate / 3:
            out.append(rng.choice([b for b in "ACGT" if b != s[i]]))
            i += 1
        elif r < 2 * rate / 3:
            out.append(rng.choice("ACGT") if not hp_bias else s[i])
        elif r < rate:
            i += 1
        else:
            out.append(s[i])
            i += 1
    return "".join(out)


def window_case(rng, n, k, rate, drift=0, pad_left=0, pad_right=0, repeat_unit=0):
    """x = read window; y = the other read's window at offset -k .. n+k, as verify_window builds it."""
    flank = 80
    if repeat_unit:
        unit = rand_seq(rng, repeat_unit)
        core = (unit * ((n + 2 * flank) // repeat_unit + 2))[: n + 2 * flank]
    else:
        core = rand_seq(rng, n + 2 * flank)
    x = core[flank : flank + n]
    other = mutate(rng, core, rate)
    ys = flank - k + drift
    y = other[ys : ys + n + 2 * k]
    y = y + rand_seq(rng, n + 2 * k - len(y)) if len(y) < n + 2 * k else y
    if pad_left:
        y = "N" * pad_left + y[pad_left:]
    if pad_right:
        y = y[: len(y) - pad_right] + "N" * pad_right
    return x, y


def gen_bpm_cases(seed, count):
    rng = random.Random(seed)
    cases = []
    for i in range(count):
        mode = i % 10
        if mode < 4:
            n, k = 375, 15
        elif mode < 6:
            n, k = 375, 31
        elif mode < 8:
            n = rng.randint(4, 374)
            k = max(1, int(n * 0.04))
        elif mode == 8:
            n = rng.randint(25, 375)
            k = min(31, max(1, int(n * 0.04)) * 2)
        else:
            n, k = rng.randint(1, 40), rng.randint(0, 3)
        rate = rng.choice([0.0, 0.002, 0.004, 0.01, 0.02, 0.05, 0.12])
        drift = rng.choice([0, 0, 0, 1, -1, 3, -4, k, -k, k + 2])
        padl = rng.choice([0, 0, 0, rng.randint(1, max(1, k))])
        padr = rng.choice([0, 0, 0, rng.randint(1, max(1, k))])
        rep = rng.choice([0, 0, 0, 0, 1, 2, 7, 23])
        x, y = window_case(rng, n, k, rate, drift, padl, padr, rep)
        cases.append({"k": k, "x": x, "y": y})
    # SURVEY 8c known answer
    cases.append({"k": 3, "x": "ACGTACGTTGCAAGCTTAGC", "y": "NNNACGTACGTGCAAGCTTAGCANNN"})
    return cases


def main():
    if not os.path.exists(HARNESS):
        sys.exit("build the reference harness first: make -f oracle/ref.mk")
    os.makedirs(OUT, exist_ok=True)

    cases = gen_bpm_cases(20261003, 600)
    rep = ask([f"bpm {c['k']} {c['x']} {c['y']}" for c in cases])
    for c, r in zip(cases, rep):
        site, err = r.split()
        c["end_site"], c["err"] = int(site), int(err)
    json.dump({"source": "Reserve_Banded_BPM via oracle/_ref/ha14_kernels", "cases": cases},
              open(os.path.join(OUT, "bpm_k5.json"), "w"), separators=(",", ":"))
    n_hit = sum(c["err"] >= 0 for c in cases)
    print(f"bpm_k5: {len(cases)} cases, {n_hit} hits")

    cases = gen_bpm_cases(777, 400)
    rep = ask([f"path {c['k']} {c['x']} {c['y']}" for c in cases])
    for c, r in zip(cases, rep):
        f = r.split()
        c["end_site"], c["err"] = int(f[0]), int(f[1])
        if c["err"] >= 0:
            c["start_site"], c["path_len"], c["path"] = int(f[2]), int(f[3]), f[4]
            assert f[5] == "|"
            c["cigar_start"], c["cigar_end"], c["cigar_err"], c["cigar"] = int(f[6]), int(f[7]), int(f[8]), f[9]
    json.dump({"source": "Reserve_Banded_BPM_PATH + generate_cigar via oracle/_ref/ha14_kernels",
               "ops": "path digits are stored end-to-start: 0 match 1 mismatch 2 y-only 3 x-only; cigar M=0 X=1 I=2 D=3",
               "cases": cases},
              open(os.path.join(OUT, "bpm_k6.json"), "w"), separators=(",", ":"))
    print(f"bpm_k6: {len(cases)} cases, {sum(c['err'] >= 0 for c in cases)} hits")

    rng = random.Random(4242)
    seqs = []
    for i in range(24):
        n = rng.choice([60, 150, 400, 1000, 3000, 8000])
        s = rand_seq(rng, n)
        if i % 3 == 1:  # homopolymer-rich
            s = "".join(b * rng.choice([1, 1, 1, 2, 3, 6]) for b in s)[:n]
        if i % 4 == 2:  # tandem repeat block in the middle
            u = rand_seq(rng, rng.randint(20, 60))
            s = s[: n // 3] + (u * 80)[: n // 3] + s[2 * n // 3 :]
        if i % 6 == 5:
            s = s[: n // 2] + "N" + s[n // 2 + 1 :]
        seqs.append(s)
    params = [(51, 51, 1), (19, 19, 0), (10, 15, 0)]
    lines, meta = [], []
    for s in seqs:
        for (w, k, hpc) in params:
            lines.append(f"sketch {w} {k} {hpc} {s}")
            meta.append({"w": w, "k": k, "hpc": hpc, "seq": s})
    rep = ask(lines)
    for m, r in zip(meta, rep):
        f = r.split()
        m["mz"] = [[int(v) for v in t.split(":")] for t in f[1:]]
        assert len(m["mz"]) == int(f[0])
    json.dump({"source": "ha_sketch via oracle/_ref/ha14_kernels", "fields": "hash,pos,rev,span", "cases": meta},
              open(os.path.join(OUT, "sketch.json"), "w"), separators=(",", ":"))
    print(f"sketch: {len(meta)} cases, {sum(len(m['mz']) for m in meta)} minimizers")

    rng = random.Random(99)
    kc = []
    for i in range(60):
        n = rng.randint(20, 400)
        t = rand_seq(rng, n)
        q = mutate(rng, t, rng.choice([0.0, 0.01, 0.05]))
        if i % 3 == 0 and n > 60:
            a = rng.randint(10, n - 40)
            q = q[:a] + q[a + rng.randint(1, 30):]
        if i % 3 == 1:
            a = rng.randint(5, len(q) - 5)
            q = q[:a] + rand_seq(rng, rng.randint(1, 30)) + q[a:]
        kc.append({"a": 2, "b": 4, "q": 4, "e": 2, "w": 500, "zdrop": 400, "query": q, "target": t})
    kc.append({"a": 2, "b": 4, "q": 4, "e": 2, "w": 500, "zdrop": 400,
               "query": "ACGTACGTTGCAAGCTTAGCACGTACGTTGCAAGCTTAGC", "target": "ACGTACGTTGCAAGCGCACGTACGTTGCAAGCTTAGC"})
    rep = ask([f"ksw {c['a']} {c['b']} {c['q']} {c['e']} {c['w']} {c['zdrop']} {c['query']} {c['target']}" for c in kc])
    for c, r in zip(kc, rep):
        f = r.split()
        c["score"], c["cigar"] = int(f[0]), (f[1] if len(f) > 1 else "")
    json.dump({"source": "ksw_extz2_sse via oracle/_ref/ha14_kernels", "cases": kc},
              open(os.path.join(OUT, "ksw_extz2.json"), "w"), separators=(",", ":"))
    print(f"ksw_extz2: {len(kc)} cases")


if __name__ == "__main__":
    main()
