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
