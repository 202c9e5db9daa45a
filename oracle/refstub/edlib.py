"""Stand-in for `edlib` (absent in the build container): exact global unit-cost Levenshtein,
which is what edlib.align(a, b) with default arguments (mode "NW") returns as "editDistance".
TEST INFRASTRUCTURE ONLY — see oracle/refstub/pysam.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import orc  # noqa: E402


def align(query, target, mode="NW", task="distance", k=-1, additionalEqualities=None):
    d = orc.edit_distance(query.encode() if isinstance(query, str) else query,
                          target.encode() if isinstance(target, str) else target)
    return {"editDistance": d, "alphabetLength": 4, "locations": [(None, len(target) - 1)], "cigar": None}
