"""The CPU restatement of K5/K6 (oracle/bpm.c) against vectors minted from the
reference's own Reserve_Banded_BPM / _PATH / generate_cigar (tools/make_golden_kernels.py)."""
import json
import os

from tests import oracle_lib as O


def test_k5_matches_reference_golden(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "bpm_k5.json")))["cases"]
    assert len(cases) > 500
    for c in cases:
        site, err = O.bpm(c["x"], c["y"], c["k"])
        assert err == c["err"], c
        if err >= 0:
            assert site == c["end_site"], c


def test_extension_matches_reference_golden(golden_dir):
    """orc_bpm_extension against the reference's alignment_extension (tools/make_golden_bpm_ext.py: 400 windows, both directions, where
    an insertion / deletion / unrelated sequence begins some way in): how far the window aligns within the threshold and at what cost --
    what non_trim_error_rate charges an unmatched window (orc_asm_params.partial_charge)"""
    cases = json.load(open(os.path.join(golden_dir, "bpm_ext.json")))["cases"]
    assert len(cases) == 400
    for c in cases:
        assert O.bpm_extension(c["x"], c["y"], c["k"], c["dir"]) == (c["aligned"], c["err"], c["p_end"], c["t_end"]), c


def test_k5_survey_known_answer():
    # end_site=21 err=1
    assert O.bpm("ACGTACGTTGCAAGCTTAGC", "NNNACGTACGTGCAAGCTTAGCANNN", 3) == (21, 1)


def test_k6_path_and_cigar_match_reference_golden(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "bpm_k6.json")))["cases"]
    hits = 0
    for c in cases:
        site, err, start, path = O.bpm_path(c["x"], c["y"], c["k"])
        assert err == c["err"], c
        if err < 0:
            continue
        hits += 1
        assert site == c["end_site"] and start == c["start_site"]
        assert "".join(str(b) for b in path) == c["path"]
        st, en, er, cg = O.generate_cigar(path, c["x"], c["y"], start, site, err)
        assert (st, en, er, cg) == (c["cigar_start"], c["cigar_end"], c["cigar_err"], c["cigar"]), c
    assert hits > 200


def _noisy_cases(seed, count, ks, rates=(0.0, 0.02, 0.1, 0.2, 0.3)):
    import random
    rng = random.Random(seed)
    out = []
    for it in range(count):
        n = rng.choice([375, 375, 375, 200, 64, 17])
        k = rng.choice(ks) if n >= 200 else rng.choice([1, 3, 8])
        ref = "".join(rng.choice("ACGT") for _ in range(n + 2 * k + 240))
        rate = rng.choice(rates)
        x = []
        for c in ref[110:110 + n + 60]:
            r = rng.random()
            if r < rate / 3: x.append(rng.choice("ACGT"))
            elif r < 2 * rate / 3: x.append(c); x.append(rng.choice("ACGT"))
            elif r < rate: pass
            else: x.append(c)
        x = "".join(x)[:n]
        x += "".join(rng.choice("ACGT") for _ in range(n - len(x)))
        ys = 110 - k + rng.randint(-min(k, 10), min(k, 10))
        out.append({"k": k, "x": x, "y": ref[ys:ys + n + 2 * k]})
    return out


def test_wide_bands_against_plain_dp_and_the_64_bit_code():
    """orc_bpm_wide / orc_bpm_path_wide (bands up to 255 rows; the reference stops at 63): equal to the 64-bit functions wherever
    both apply, equal in distance and end row to a plain banded DP, and every wide path is a valid alignment of its distance"""
    n_wide = n_both = 0
    for c in _noisy_cases(5, 1500, [3, 15, 31, 40, 63, 94]):
        x, y, k, n = c["x"], c["y"], c["k"], len(c["x"])
        site, err = O.bpm_wide(x, y, k)
        best, ends = O.banded_dp_plain(x, y, k)
        if best <= k:
            assert err == best and (site - (n - 1)) in ends, c
        else:
            assert err == -1, c
        if k <= 31:
            assert O.bpm(x, y, k) == (site, err)
            if err >= 0:
                assert O.bpm_path(x, y, k) == O.bpm_path(x, y, k, wide=True)
                n_both += 1
        elif err >= 0:
            s2, e2, start, path = O.bpm_path(x, y, k, wide=True)
            assert (s2, e2) == (site, err)
            xi, yi, e = 0, start, 0
            for op in path[::-1]:
                if op == 0:
                    assert x[xi] == y[yi]; xi += 1; yi += 1
                elif op == 1:
                    assert x[xi] != y[yi]; e += 1; xi += 1; yi += 1
                elif op == 2:
                    e += 1; yi += 1
                else:
                    e += 1; xi += 1
            assert (xi, yi - 1, e) == (n, site, err), c
            n_wide += 1
    assert n_wide > 300 and n_both > 300
