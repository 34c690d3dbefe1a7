"""The gather kernels are worth what they keep in flight -- and that is decided by the compiler's scheduler, not by the
source.  tools/isa_check.py reads the gfx950 assembly of the engine (hipcc cross-compiles without a GPU) and checks, per
kernel, that a step's row loads are requested TOGETHER (runs of consecutive 16-byte loads between two waits) and that
nothing spills: round 5 found two silent regressions of exactly this kind (profiles/r05_walk_occupancy_hint_ab.txt)."""
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_gather_kernels_keep_their_loads_in_flight():
    import isa_check
    ks, bad = isa_check.check(rebuild=True)
    assert len(ks) > 150                      # the assembly was found and parsed
    for family in ("and_walk_kernel", "and_band_walk_kernel", "and_screen_kernel", "and_refine_kernel", "count_walk_kernel", "count_screen_kernel", "count_refine_kernel"):
        assert any(k[0] == family for k in ks), family
    assert bad == [], "\n".join(bad)
