"""The bench line committed under profiles/ (produced by `python bench.py` on the GPU box) carries every key the
driver contract names.  CPU-only: it reads the committed evidence, it does not run the bench."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_default_bench_line_has_the_contract_keys():
    d = _line("r01_bench_batch1.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "frames/s" and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["steps"] * d["config"]["batch_per_gpu"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["us_per_launch"] * 1e-6) / 1e9) < 0.01 * r["achieved"]
    assert r["traffic"] is not None and 1.0 <= r["traffic"] / r["bytes_per_launch"] < 1.1      # PMC traffic ~ algorithmic bytes
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]
    assert 0 < d["prefill"]["mfma_frac"] < 1 and d["prefill"]["peak_tflops_bf16_dense"] == 2500.0


def test_other_config_lines():
    for name in ("r01_bench_batch8.json", "r01_bench_batch8_pruned50.json"):
        d = _line(name)
        assert d["config"]["batch_per_gpu"] == 8 and d["value"] > 0 and d["roofline"]["traffic"] is None
        assert d.get("cpu_baseline") is None             # the CPU leg runs in the default (batch 1, N = 1) invocation only


def test_default_batch_follows_baseline_configs():
    """N = 1: BASELINE configs[1] (batch 1); N > 1: configs[4], 64 utterances over the N GPUs"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.default_batch(1) == 1
    assert [bench.default_batch(n) for n in (2, 4, 8)] == [32, 16, 8]
    assert bench.MIXED_L == [32, 64, 96, 128, 192, 256, 384, 512] and sum(bench.MIXED_L) == 1664
