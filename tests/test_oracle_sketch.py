"""oracle/sketch.c against minimizers minted from the reference's ha_sketch."""
import json
import os

from tests import oracle_lib as O


def test_sketch_matches_reference_golden(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "sketch.json")))["cases"]
    total = 0
    for c in cases:
        got = O.sketch(c["seq"], c["w"], c["k"], c["hpc"])
        exp = c["mz"]
        assert len(got) == len(exp), (c["w"], c["k"], c["hpc"], len(c["seq"]))
        for g, e in zip(got, exp):
            assert [int(g["hash"]), int(g["pos"]), int(g["rev"]), int(g["span"])] == e
        total += len(exp)
    assert total > 10000
