"""The bench line committed under profiles/ (produced by `python bench.py` on the GPU box) carries every key the
driver contract names.  CPU-only: it reads the committed evidence, it does not run the bench."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_default_bench_line_has_the_contract_keys():
    d = _line("r03_bench_default.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "configs"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "frames/s" and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["steps"] * d["config"]["batch_per_gpu"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["us_per_launch"] * 1e-6) / 1e9) < 0.01 * r["achieved"]
    assert r["kernel_us"] <= r["us_per_launch"] and r["launches_per_step"] >= 1
    # the dominant kernel by time heads the per-kernel table, whose shares cover the step
    rows = d["roofline_by_kernel"]
    assert rows[0]["kernel"] == r["kernel"] and all(rows[i]["share_of_step_time"] >= rows[i + 1]["share_of_step_time"] for i in range(len(rows) - 1))
    assert 0.97 <= sum(x["share_of_step_time"] for x in rows) <= 1.03
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]
    assert 0 < d["prefill"]["mfma_frac"] < 1 and d["prefill"]["peak_tflops_bf16_dense"] == 2500.0
    assert abs(d["prefill"]["mfma_frac_issued"] - 3 * d["prefill"]["mfma_frac"]) < 2e-3
    # every single-GPU BASELINE configuration rides in the same line, at 1024 steps
    cf = d["configs"]
    assert set(cf) == {"batch1_bf16kv", "batch1_f32kv", "batch1_bf16x2kv", "batch8_mixed_bf16kv", "batch8_mixed_f32kv", "batch8_mixed_bf16x2kv",
                       "batch64_mixed_bf16kv", "pruned50_batch8_bf16kv", "pruned50_batch1_bf16kv"}
    for k, v in cf.items():
        assert v["steps"] == (512 if k.startswith("batch64") else 1024) and v["frames_per_s"] > 0 and 0 < v["step_frac_of_hbm_peak"] < 1
    # the N = 1 point of BASELINE configs[4] (64 utterances on one GPU) is in the driver-run line
    assert "configs[4]" in cf["batch64_mixed_bf16kv"]["workload"] and cf["batch64_mixed_bf16kv"]["frames_per_s"] > cf["batch8_mixed_bf16kv"]["frames_per_s"]
    assert cf["pruned50_batch8_bf16kv"]["decode_weight_bytes"] < 0.45 * cf["batch8_mixed_bf16kv"]["decode_weight_bytes"]


def test_traffic_file_names_the_kernels_of_the_bench_line():
    """profiles/traffic.json (PMC passes) is keyed by the kernel names bench.py reports; HBM traffic of the weight streams ~ algorithmic bytes"""
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["batch1"]
    d = _line("r03_bench_default.json")
    by = {x["kernel"]: x for x in d["roofline_by_kernel"]}
    gemvs = [n for n in by if n.startswith("k_gemv_small<")]           # (qkv/o/cq/co, wi, wo, logits: the weight streams of a batch-1 step)
    assert len(gemvs) >= 3
    for name in gemvs:
        assert name in t, name
        assert 0.95 <= t[name]["hbm_bytes_per_launch"] / by[name]["bytes_per_launch"] <= 1.15, name
    # (the bench line carries the traffic file that was current when it ran; the PMC passes that follow it rewrite the file, and the
    # per-launch means move by ~0.1 % between passes)
    assert abs(d["roofline"]["traffic"] - t[d["roofline"]["kernel"]]["hbm_bytes_per_launch"]) <= 5e-3 * d["roofline"]["traffic"]


def test_other_config_lines():
    for name in ("r03_bench_batch8.json", "r03_bench_batch8_pruned50.json"):
        d = _line(name)
        assert d["config"]["batch_per_gpu"] == 8 and d["value"] > 0 and d["roofline"]["kernel"]
        assert d.get("cpu_baseline") is None             # the CPU leg runs in the default (batch 1, N = 1) invocation only
        assert "text bytes [32, 64, 96, 128, 192, 256, 384, 512]" in d["config"]["workload"]


def test_default_batch_follows_baseline_configs():
    """N = 1: BASELINE configs[1] (batch 1); N > 1: configs[4], 64 utterances over the N GPUs"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.default_batch(1) == 1
    assert [bench.default_batch(n) for n in (2, 4, 8)] == [32, 16, 8]
    assert bench.MIXED_L == [32, 64, 96, 128, 192, 256, 384, 512] and sum(bench.MIXED_L) == 1664
