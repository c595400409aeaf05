"""The committed bench line (profiles/r01_bench.json, produced by bench.py on an MI355X) carries every field of the
driver contract, the roofline block and the CPU baseline block, with consistent arithmetic."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_schema_and_arithmetic():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench.json")))
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str),
                     ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert "vs_baseline" in d and d["vs_baseline"] is None            # BASELINE.md publishes no number for this metric
    assert d["unit"] == "maps/s" and d["scaling"] == "weak" and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    batch = d["config"]["global_batch"] // d["n_gpus"]
    assert abs(d["value"] - d["n_gpus"] * batch / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["bytes_per_launch"] / r["us_per_launch"] / 1e3) / r["achieved"] < 1e-3
    assert r["traffic"] is None or r["traffic"] >= 0.95 * r["bytes_per_launch"]     # measured HBM bytes cover the algorithmic ones
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
