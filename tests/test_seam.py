"""CPU: the seam harness — the oracle's kernels slotted behind the REFERENCE's own operator tables must
leave its .265 byte-identical.  The committed report (tests/golden/seam_report.json) is always checked; where the reference binary is
present (the builder container) one configuration is re-run live."""
from __future__ import annotations

import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_seam_report_is_byte_identical():
    rep = json.load(open(os.path.join(ROOT, "tests", "golden", "seam_report.json")))
    assert len(rep["runs"]) >= 3
    for r in rep["runs"]:
        assert r["identical"] and r["md5_plain"] == r["md5_seam"], r["config"]
    slow = [r for r in rep["runs"] if "slow" in r["config"]][0]["oracle_calls"]
    for fam in ("sad", "sad4", "sad3", "sse", "had", "fwd_transform", "quant", "inv_transform", "residual", "deblock_luma", "deblock_chroma",
                "interp", "sao_bo", "sao_stats"):
        assert slow[fam] > 0, fam          # every patched family was really exercised


@pytest.mark.skipif(not os.path.exists("/root/reference/ubuntu_x64/appencoder"), reason="reference binary only exists in the builder container")
def test_seam_live_one_config():
    sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_probe"))
    import seam_harness
    saved = seam_harness.CONFIGS
    try:
        seam_harness.CONFIGS = [dict(name="208x120 slow qp27", w=208, h=120, frames=4, seed=3, abc=(17, 23, 9),
                                     args=["-preset", "slow", "-rc", "0", "-qp", "27", "-iper", "128"])]
        rep = seam_harness.run()
    finally:
        seam_harness.CONFIGS = saved
    assert rep["runs"][0]["identical"] and rep["runs"][0]["oracle_calls"]["had"] > 0
