"""The verification switches (voldor_amd/csrc/vk_debug.h): every switch the header documents is known to the library, takes the values the header lists for it (default first)
and refuses others; an unknown name is refused.  Host logic only -- no device is touched."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _documented():
    text = open(os.path.join(ROOT, "voldor_amd", "csrc", "vk_debug.h")).read()
    head = text[text.index("/* name / values (default first):"):text.index("int vk_debug_switch(")]
    out = {}
    for m in re.finditer(r'^ \*   "(\w+)"\s+((?:\d+|n > 0)(?: \| (?:\d+|n > 0))*)', head, re.M):
        out[m.group(1)] = [v.strip() for v in m.group(2).split("|")]
    return out


def test_every_documented_switch_is_known_and_takes_its_documented_values():
    from voldor_amd import capi
    L = capi.lib()
    L.vk_debug_switch.restype = C.c_int
    L.vk_debug_switch.argtypes = [C.c_char_p, C.c_int]
    doc = _documented()
    assert len(doc) >= 16 and "strict_filter" in doc and "strict_table_filter" in doc and "fb_ride" in doc, sorted(doc)
    for name, values in doc.items():
        default = int(values[0])
        try:
            for v in values:
                v = 7 if v == "n > 0" else int(v)
                assert L.vk_debug_switch(name.encode(), v) >= 0, (name, v)
            if "n > 0" not in values:  # a plain on / off switch folds any value to 0 | 1; the others refuse what is not listed
                folded = sorted(int(v) for v in values) == [0, 1]
                assert (L.vk_debug_switch(name.encode(), 3) >= 0) == folded, name
        finally:
            assert L.vk_debug_switch(name.encode(), default) >= 0, name
    assert L.vk_debug_switch(b"no_such_switch", 1) == -1
