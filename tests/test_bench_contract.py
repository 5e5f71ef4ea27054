"""bench.py prints ONE JSON line with the fields of the task contract plus `roofline` and `cpu_baseline`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-seconds", "1", "--extras", "0"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["higher_is_better"] is True and j["vs_baseline"] is None
    assert j["unit"] == "Mrays/s" and j["dtype"] == "f32" and j["scaling"] == "strong" and j["data"].startswith("synthetic")
    assert j["config"]["workload"].startswith("config2") and "model" not in j["config"]
    assert j["value"] > 1000 and abs(j["value"] - j["config"]["rays_per_step"] / j["ms_per_step"] / 1e3) / j["value"] < 1e-3
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # the fused kernel is VALU-bound: frac is the VALU issue fraction of the timed kernel and must be a fraction
    assert r["bound"] == "valu" and r["pmc_source"]["kind"] in ("measured in this run", "file")
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    h = r["hbm"]
    assert h["peak"] == 8000.0 and h["unit"] == "GB/s" and 0.0 < h["frac"] <= 1.0 and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-3
    assert abs(r["traffic"] - h["read_bytes"] - h["write_bytes"]) <= 2 and r["traffic"] >= 0.5 * h["compulsory_bytes"]
    assert "algorithmic_bytes_per_launch" in r and 0.0 < r["lane_utilisation"] <= 1.0
    # what frac hides: idle lanes, and the quarter-rate instructions (the second only where this rocprofv3 has the counter)
    assert abs(r["useful_lane_frac"] - r["frac"] * r["lane_utilisation"]) < 2e-4
    assert "frac_issue_cycles" in r and (r["frac_issue_cycles"] is None or r["frac"] <= r["frac_issue_cycles"] <= 1.0)
    assert j["extra"]["frame_event_ms_min"] <= j["extra"]["frame_event_ms_median"]
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == "Mrays/s" and c["value"] > 0 and c["cores"] >= 1


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode != 0 and "no CPU path" in (out.stderr + out.stdout)


def test_counting_kernels_are_told_from_production_ones_by_their_count_argument():
    """bench.py prices the PMC counters of the production kernels only.  COUNT is the first template argument of the shade kernel
    and the second of every other one; a trailing `true>` may be EXISTS or ALT."""
    sys.path.insert(0, ROOT)
    import bench
    cases = {"hrt_wf_walk_closest_kernel<0, false, false, true>": False, "hrt_wf_walk_closest_kernel<0, false, true, true>": False,
             "hrt_wf_walk_shadow_kernel<0, false, true>": False, "hrt_wf_walk_shadow_kernel<0, true, false>": True,
             "hrt_wf_shade_kernel<true, false>": True, "hrt_wf_shade_kernel<false, true>": False, "hrt_wf_finish_kernel<1, false>": False,
             "hrt_path_trace_kernel<hrt::TracerPackedT<0>, true>": True, "hrt_path_trace_kernel<hrt::TracerFlat, false>": False,
             "hrt_wf_resolve_kernel": False}
    for name, want in cases.items():
        assert bench.is_counting_kernel(name) == want, name


def test_roofline_scales_child_counts_and_prices_a_tile_against_one_gpu():
    """The extras' PMC child renders fewer samples than the timed frame (counters are linear in spp), and at N > 1 the counts are
    those of rank 0's tile: one GPU's peak, the tile's share of the compulsory bytes -- never N x the peak."""
    sys.path.insert(0, ROOT)
    import bench
    from ilgpu_raytracing_amd import scenes
    res = {"cfg": scenes.CONFIGS[4], "fused": False, "path_ms": 100.0, "pixels": 3840 * 2160, "alg_bytes": 8e9, "rays": 1e9}
    stage = {"SQ_INSTS_VALU": 1e9, "SQ_ACTIVE_INST_VALU": 1e9, "SQ_THREAD_CYCLES_VALU": 32e9, "SQ_WAVE_CYCLES": 4e9, "SQ_WAIT_ANY": 2e9,
             "hbm_read_bytes": 1e9, "hbm_write_bytes": 5e8, "SQ_INSTS_VALU_TRANS_F32": 1e8, "kernels": ["k"]}
    a = bench.roofline_of(res, {"stage": stage}, {"kind": "measured in this run", "child_spp": 8}, 1, count_scale=8.0)
    assert a["valu_insts_per_launch"] == 8e9 and a["counts_scaled_by"] == 8.0 and a["traffic"] == 12e9
    assert abs(a["frac"] - 8e9 / 0.1 / 1e9 / bench.VALU_PEAK_GINST) < 1e-4 and abs(a["lane_utilisation"] - 0.5) < 1e-9
    assert abs(a["useful_lane_frac"] - a["frac"] * 0.5) < 1e-4 and abs(a["frac_issue_cycles"] - a["frac"] * 1.3) < 1e-3
    b = bench.roofline_of(res, {"stage": stage}, {"kind": "measured in this run"}, 1, tile_of=8)
    assert b["peak"] == round(bench.VALU_PEAK_GINST, 1) and "tile" in b["priced"]
    assert b["hbm"]["compulsory_bytes"] == int(3840 * 2160 * 104 / 8) and b["algorithmic_bytes_per_launch"] == int(1e9)
    c = bench.roofline_of(res, None, {"kind": "none", "reason": "x"}, 1, tile_of=8)
    assert c["frac"] is None and c["achieved"] is None
