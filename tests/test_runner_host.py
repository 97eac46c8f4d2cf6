"""Host side of scripts/run_flownet_many.py (no GPU): grouping of a pair list into batches of equal-size images."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    spec = importlib.util.spec_from_file_location("run_flownet_many", os.path.join(ROOT, "scripts", "run_flownet_many.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_groups_read_every_file_once_and_keep_order():
    m = _load()
    sizes = {"a": (4, 6), "b": (4, 6), "c": (4, 6), "d": (8, 6), "e": (4, 6), "f": (4, 6), "g": (4, 6), "h": (4, 6)}
    reads = []

    def read(name):
        reads.append(name)
        key = name[0]
        return np.full((1, 3) + sizes[key], ord(key) + (name[1] == "1"), np.float32)

    entries = [[k + "0", k + "1", k + ".flo"] for k in "abcdefgh"]
    groups = list(m.groups_of(entries, 3, read))
    assert [[e[2][0] for e in g[0]] for g in groups] == [["a", "b", "c"], ["d"], ["e", "f", "g"], ["h"]]
    assert sorted(reads) == sorted(n for e in entries for n in e[:2]) and len(reads) == 16          # every image decoded exactly once
    for ents, i0, i1 in groups:
        assert i0.shape == (len(ents), 3) + sizes[ents[0][2][0]] and i1.shape == i0.shape
        for k, e in enumerate(ents):
            assert float(i0[k, 0, 0, 0]) == ord(e[2][0]) and float(i1[k, 0, 0, 0]) == ord(e[2][0]) + 1
    assert list(m.groups_of([], 4, read)) == []


def test_reference_argument_form_is_detected_by_the_prototxt_positional():
    """scripts/run_flownet.py: `caffemodel deployproto img0 img1 out` (run-flownet.py:12-18) vs the built-in form with options."""
    spec = importlib.util.spec_from_file_location("run_flownet", os.path.join(ROOT, "scripts", "run_flownet.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m._positionals(["--weights", "w.npz", "--gpu", "1", "a.png", "b.png", "out.flo"]) == ["a.png", "b.png", "out.flo"]
    ref = m._positionals(["m.caffemodel", "deploy.prototxt.template", "a.png", "b.png", "out.flo", "--gpu", "2", "--verbose"])
    assert ref == ["m.caffemodel", "deploy.prototxt.template", "a.png", "b.png", "out.flo"] and ref[1].endswith((".prototxt", ".template"))
