"""Every name rt_set_int / rt_set_float / rt_set_vec* accepts is documented: uniforms in include/rt_core.h, tuning knobs in
INTEGRATION.md's table and by name in the header (the reference's uniforms are ignored when unknown, so a knob that is
missing from the documents cannot be found by trying)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_settable_name_is_documented():
    src = open(os.path.join(ROOT, "raytracing_amd", "csrc", "rt_core.hip")).read()
    names = sorted(set(re.findall(r'name_is\(name, "([A-Za-z0-9_]+)"\)', src)))
    assert len(names) >= 30
    header = open(os.path.join(ROOT, "include", "rt_core.h")).read()
    integration = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing_header = [n for n in names if not re.search(r"\b%s\b" % n, header)]
    assert not missing_header, missing_header
    uniforms = {"aspect", "fov", "camPos", "camForward", "camRight", "camUp", "offset", "numLights", "hasSun", "samplesPerPixel", "maxBounces"}
    missing_table = [n for n in names if n not in uniforms and ("`%s`" % n) not in integration]
    assert not missing_table, missing_table
