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
