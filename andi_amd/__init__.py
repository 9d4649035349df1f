"""andi_amd — MI355X (gfx950) engine for andi's anchor-distance hot path.

The product is libandihip.so (HIP kernels + C-ABI, include/andi_hip.h); this
package is the thin Python mirror of that interface used by the tests and the
benchmark.
"""
from . import lib, synth  # noqa: F401
from .lib import (AndiHipError, Context, bootstrap, Esa, Queries, M_ANI, M_JC, M_KIMURA, M_LOGDET, M_RAW,  # noqa: F401
                  dist_matrix, estimate, format_distances, match_positions, scan_rows, subject_prepare,
                  suffix_array)
