"""bench.py's host-side helpers (no GPU): the core count the CPU baselines may use, and the reduction of the committed PMC traffic
file to bytes per launch of a kernel group."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_host_cores_is_positive_and_within_the_affinity_mask():
    b = _bench()
    n = b.host_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_profiled_traffic_averages_template_instances_and_sums_kernels():
    # a step launches ONE instantiation of the templated traversal (the walker alternates): the two entries are averaged, the
    # other kernels of the group added; a file collected at another batch is not used
    b = _bench()
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_mccfr_hbm_traffic.json"))
    assert files, "the PMC reduction is a committed artifact"
    doc = json.load(open(os.path.join(ROOT, "profiles", files[-1])))
    total, src = b.profiled_traffic(("traverse", doc["update"]), doc["batch"])
    assert src == files[-1] and total is not None
    inst = [v["hbm_bytes_per_launch"] for k, v in doc["kernels"].items() if "k_traverse" in k]
    prep = [v["hbm_bytes_per_launch"] for k, v in doc["kernels"].items() if "k_prepare_infos" in k]
    assert abs(total - (sum(inst) / len(inst) + sum(prep))) < 1.0
    none, _ = b.profiled_traffic(("traverse", doc["update"]), doc["batch"] + 1)
    assert none is None


def test_strong_scaling_projection_runs_a_child_and_labels_its_result(monkeypatch):
    # the 8-GPU figure of the default line is a projection: one rank's share measured by a child `bench.py --force-sharded` (so that
    # a hung collective cannot take the contract line along), the wire assumed and said so
    import argparse
    import subprocess

    b = _bench()
    seen = {}

    def fake_run(cmd, env=None, capture_output=None, text=None, timeout=None):
        seen["cmd"], seen["env"], seen["timeout"] = cmd, env, timeout
        out = "banner\n" + json.dumps({"metric": "mccfr_infoset_updates_per_sec", "ms_per_step": 0.125, "n_gpus": 1}) + "\n"
        return subprocess.CompletedProcess(cmd, 0, stdout=out, stderr="")

    monkeypatch.setattr(subprocess, "run", fake_run)
    args = argparse.Namespace(batch=1 << 23, steps=20, warmup=5, game="leduc", regret="linear", weight="linear", sampling="external",
                              seed=7, dist_backend="nccl")
    p = b.strong_scaling_projection(args, 0, ms_single=0.75)
    assert "projection" in p["kind"] and p["ranks"] == 8 and p["batch_per_rank"] == 1 << 20
    assert abs(p["projected_speedup"] - 0.75 / (0.125 + 0.020 / 4)) < 1e-12
    cmd = seen["cmd"]
    assert "--force-sharded" in cmd and "--no-extras" in cmd and cmd[cmd.index("--batch") + 1] == str(1 << 20)
    assert seen["env"]["WORLD_SIZE"] == "1" and seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["timeout"]

    def failing_run(cmd, **kw):
        return subprocess.CompletedProcess(cmd, 3, stdout="", stderr="no device")

    monkeypatch.setattr(subprocess, "run", failing_run)
    try:
        b.strong_scaling_projection(args, 0, ms_single=0.75)
        raise AssertionError("a failed child must raise (the caller records the error in the line)")
    except RuntimeError as exc:
        assert "child exited 3" in str(exc)


def test_the_compact_line_keeps_the_contract_keys_and_ends_with_the_kmeans_numbers():
    # the driver keeps the END of stdout: the default line is a view of the detail object that fits its window (8 KB) with both
    # arithmetics' end-to-end seconds inside the last 2 KB; nothing in it is computed, only selected
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")))
    km = full["kmeans"]
    km["contract_arithmetic"] = {k: km[k] for k in ("libm", "rng", "create_s", "kmeanspp_s", "init_bounds_s", "elkan_total_s", "lookup_s",
                                                    "end_to_end_s", "points_per_s", "rms", "kernels_ms", "roofline_sinkhorn", "roofline_mfma")}
    km["contract_arithmetic"].update({"picks_differing": 232, "buckets_differing": 1180039, "adjusted_rand_index": 0.5, "matched_label_fraction": 0.6})
    line = b.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 7800
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[k] == full[k]
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    assert list(line)[-1] == "kmeans" and "per_iteration" not in text
    tail = text[-2000:]
    assert '"end_to_end_s_by_arithmetic"' in tail and '"end_to_end_s"' in tail and '"value"' in tail
    assert line["kmeans"]["contract_arithmetic"]["adjusted_rand_index"] == 0.5


def test_partition_agreement_ignores_label_order():
    import numpy as np

    b = _bench()
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, 50_000)
    same = b.partition_agreement(a, rng.permutation(256)[a], 256)
    assert abs(same["adjusted_rand_index"] - 1.0) < 1e-12 and same["matched_label_fraction"] == 1.0
    other = b.partition_agreement(a, rng.integers(0, 256, 50_000), 256)
    assert abs(other["adjusted_rand_index"]) < 0.01 and other["matched_label_fraction"] < 0.05
    half = a.copy()
    half[:25_000] = rng.integers(0, 256, 25_000)
    mid = b.partition_agreement(a, half, 256)
    assert 0.2 < mid["adjusted_rand_index"] < 0.3 and 0.49 < mid["matched_label_fraction"] < 0.52
