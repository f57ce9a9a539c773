"""CPU suite: the measurement plumbing of bench.py that needs no GPU -- the committed PMC summaries under profiles/ are read by the
same code the bench line's `roofline.traffic` comes from (the field was null because the committed file had another
layout than the reader expected)."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_traffic_of_the_step_kernels_is_read_from_the_committed_pmc_summary():
    for name in ("fused_fwd_head_dx_kernel<256, 2, 512>", "dw_table_kernel"):
        t, src = bench.pmc_traffic(name)
        assert src is not None and os.path.exists(os.path.join(ROOT, "profiles", src))
        assert t is not None and 1e5 < t < 1e9, (name, t)          # bytes per launch: megabytes, not kilobytes or counts
    assert bench.pmc_traffic("no_such_kernel") == (None, None)


def test_both_summary_layouts_are_understood(tmp_path):
    raw = {"k_raw": {"FETCH_SIZE": [100.0, 5, 120.0], "WRITE_SIZE": [50.0, 5, 60.0]}}
    cur = {"note": "x", "kernels": {"k_cur": {"traffic_bytes": 12345.0}}}
    (tmp_path / "r01_pmc.json").write_text(json.dumps(raw))
    (tmp_path / "r02_pmc.json").write_text(json.dumps(cur))
    assert bench.pmc_traffic("k_raw<1>", str(tmp_path)) == ((2 * 100.0 + 50.0) * 1024.0, "r01_pmc.json")      # FETCH_SIZE doubled (gfx950)
    assert bench.pmc_traffic("hl::k_cur", str(tmp_path)) == (12345.0, "r02_pmc.json")
    # the newest round that lists the kernel wins
    (tmp_path / "r03_pmc.json").write_text(json.dumps({"k_raw": {"FETCH_SIZE": [1.0, 1, 1.0], "WRITE_SIZE": [2.0, 1, 2.0]}}))
    assert bench.pmc_traffic("k_raw", str(tmp_path))[1] == "r03_pmc.json"
