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
