"""bench.py's cpu_baseline leg (CPU only): the port timed on a bounded sample, the thread count measured, and the comparison with
the one recorded timing of the reference's own code."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_baseline_leg_is_labelled_and_measured():
    sys.path.insert(0, ROOT)
    import bench
    c = bench.cpu_baseline(seconds_budget=2.0, nbath=11)
    assert c["kind"] == "port" and c["unit"] == "GB/s" and c["value"] > 0
    assert c["cores"] in (1, c["threads_probe"]["chosen"])
    assert c["threads_probe"]["affinity_mask_cores"] >= 1 and str(c["threads_probe"]["chosen"]) in c["threads_probe"]["s_per_c2_product_by_threads"]
    assert "NOT the reference's own code" in c["value_is"]
    r = c["c2_port_vs_reference_code"]
    assert r["reference_code_s_per_hxv_recorded"] == 0.272 and r["port_speedup_over_reference_code"] > 1.0
