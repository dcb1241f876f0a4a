"""Host-side SV extraction from contig alignments: the part of FocalSV's step 4 that stays Python
(focalsv/4_sv_calling/Dippav/*.py).  Restated here against golden vectors captured from the reference
modules (tests/golden/dippav_*.json, tools/make_golden_dippav.py)."""
from .signatures import (AlignedSegment, PROFILES, cluster_del, cluster_ins, extract_sig_from_cigar, extract_sig_from_split,  # noqa: F401
                         merge_all, pair_sig, signatures_one_hap, sort_sig)
