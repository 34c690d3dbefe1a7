"""The measurement tables in the documents are GENERATED from the files under profiles/ (tools/render_tables.py): the
copies in the tree must be what the generator makes of those files now, and DESIGN.md must not carry a hand-typed kernel
time outside its generated block."""
import os
import re
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_generated_tables_are_current():
    import render_tables as rt
    assert open(os.path.join(ROOT, "profiles", "TABLES.md")).read() == rt.render_tables()
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    b, e = design.index(rt.DESIGN_BEGIN) + len(rt.DESIGN_BEGIN), design.index(rt.DESIGN_END)
    assert design[b:e] == rt.design_block()
    import json
    idx = json.load(open(os.path.join(ROOT, "profiles", "INDEX.json")))
    assert open(os.path.join(ROOT, "profiles", "README.md")).read() == rt.render_readme(idx)
    assert "Not described in INDEX.json" not in rt.render_readme(idx)


def test_design_md_has_no_hand_typed_kernel_time():
    import render_tables as rt
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    outside = design[:design.index(rt.DESIGN_BEGIN)] + design[design.index(rt.DESIGN_END):]
    assert re.findall(r"\d[\d.]*\s?ms\b", outside) == []
