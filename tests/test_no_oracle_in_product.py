"""The oracle is test infrastructure: nothing under cirkit_amd/ may import or call it, and the
product must not reach for the reference either."""
import os
import re

from conftest import ROOT


def test_product_never_touches_oracle_or_reference():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cirkit_amd")):
        for f in files:
            if not f.endswith((".py", ".hip", ".h", ".cpp")):
                continue
            text = open(os.path.join(dirpath, f), encoding="utf-8").read()
            if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "torch_oracle" in text:
                bad.append(os.path.join(dirpath, f))
            if "/root/reference" in text:
                bad.append(os.path.join(dirpath, f) + " (reads /root/reference)")
    assert not bad, bad


def test_gpu_entry_points_do_not_read_the_reference_checkout():
    for f in ("bench.py", "__graft_entry__.py"):
        assert "/root/reference" not in open(os.path.join(ROOT, f), encoding="utf-8").read()
