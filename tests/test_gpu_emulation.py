"""BASELINE configs[2]/[3] emulated on a 24-fragment subset (tools/emulate_3dmatch.py): synthetic 3DMatch-layout tree ->
generate_desc (two ranks sharing the GPU over gloo) -> evaluate on the 3DMatch-like and 3DLoMatch-like pair lists ->
the CPU oracle re-derives seeded pairs from the same files.  The full-size run (433 fragments, 1623 + 1781 pairs) is
profiles/r02_config3_emulation.json."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_emulated_3dmatch_subset(tmp_path, capsys):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import json
    import emulate_3dmatch as E
    out = tmp_path / "summary.json"
    assert E.main(["--work", str(tmp_path / "emu"), "--fragments", "24", "--scenes", "2", "--ranks", "2", "--oracle-pairs", "5",
                   "--keypoints", "1500", "--out", str(out)]) == 0
    s = json.load(open(out))
    assert s["fragments"] >= 24 and s["ranks"] == 2
    for bench in ("3DMatch", "3DLoMatch"):
        b = s["benchmarks"][bench]
        assert b["pairs"] > 0 and 0.0 <= b["FMR@0.05"] <= 1.0 and 0.0 <= b["registration_recall"] <= 1.0
        o = b["oracle_agreement"]
        assert o["inlier_counts_identical"] and o["max_inlier_ratio_difference"] < 1e-8
        assert o["registration_decisions_equal"] >= o["pairs_checked"] - 1      # a borderline RANSAC tie may differ
    # descriptor files of every fragment exist, written by either rank
    n = sum(len([f for f in files if f.endswith(".npz")]) for _, _, files in os.walk(tmp_path / "emu" / "desc"))
    assert n == s["fragments"]
    # the sharded run reproduces the single-process run bit for bit (files and metrics)
    import numpy as np
    out1 = tmp_path / "summary1.json"
    assert E.main(["--work", str(tmp_path / "emu1"), "--fragments", "24", "--scenes", "2", "--ranks", "1", "--oracle-pairs", "0",
                   "--keypoints", "1500", "--out", str(out1)]) == 0
    s1 = json.load(open(out1))
    for root, _, files in os.walk(tmp_path / "emu" / "desc"):
        for f in files:
            a = np.load(os.path.join(root, f))
            b = np.load(os.path.join(str(root).replace(str(tmp_path / "emu"), str(tmp_path / "emu1")), f))
            assert all((a[k] == b[k]).all() for k in ("points", "xyz", "feature")), f
    for bench in ("3DMatch", "3DLoMatch"):
        for key in ("pairs", "FMR@0.05", "FMR@0.20", "registration_recall"):
            assert s["benchmarks"][bench][key] == s1["benchmarks"][bench][key], (bench, key)
