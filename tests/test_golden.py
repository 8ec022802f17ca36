"""Committed golden vectors (tests/golden/*.json, made by scripts/make_golden.py from the CPU oracle).

CPU tests: the oracle still reproduces them (freezes rp_math.h, the RNG definitions and the f32 operation order
across rounds).  GPU tests: the HIP path reproduces the same committed bits without the oracle in the loop.
"""
import json
import os

import numpy as np
import pytest

import oracle
from lloyd_fixtures import flop_hist, flop_like_points, flop_metric, smooth_metric, turn_like_points
from robopoker_amd import Game

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32).tolist()


def _hyper(case):
    hp = oracle.default_hyper()
    hp.prune_warmup = case["prune_warmup"]
    hp.prune_threshold = case["prune_threshold"]
    return hp


def _check_tables(rows, case):
    assert bits(rows["regret"]) == case["regret_bits"]
    assert bits(rows["weight"]) == case["weight_bits"]
    assert bits(rows["payoff"]) == case["payoff_bits"]
    assert rows["visits"].tolist() == case["visits"]


def _points(case):
    if case["kind"] == "sinkhorn":
        return (flop_like_points(case["N"], bins=case["bins"], mass=case["mass"], seed=case["seed"]),
                smooth_metric(case["bins"], case["seed"]))
    return turn_like_points(case["N"], bins=case["bins"], mass=case["mass"], seed=case["seed"]), None


def _run_kmeans(km, case):
    assert km.init_centroids().tolist() == case["chosen"]
    km.init_bounds()
    for k in range(3):
        d, sizes, _ = km.step()
        assert bits(d) == case["drift_bits"][k]
    assert sizes.tolist() == case["sizes"]
    b, dist = km.assign() if hasattr(km, "assign") else km.lookup()
    assert b.tolist() == case["buckets"] and bits(dist) == case["distance_bits"]
    c, w = km.centroids()
    assert w.tolist() == case["centroid_weight"] and int(c.astype(np.uint64).sum()) == case["centroid_checksum"]
    assert bits(km.metric()) == case["metric_bits"]
    assert bits([km.rms()])[0] == case["rms_bits"]


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("case", load("mccfr_tables.json"), ids=lambda c: f"{c['game']}-{c['regret']}-{c['sampling']}")
def test_oracle_reproduces_mccfr_golden(case):
    s = oracle.OracleSolver(Game(case["game"]), case["regret"], case["weight"], case["sampling"], batch=case["batch"],
                            seed=case["seed"], hyper=_hyper(case))
    for _ in range(case["steps"]):
        s.step()
    _check_tables(s.export(), case)
    assert list(s.counters()) == case["counters"]
    assert bits([s.exploitability()])[0] == case["exploitability_bits"]


def test_oracle_reproduces_sinkhorn_golden():
    g = load("sinkhorn.json")
    tri = flop_metric()
    for c in g["sinkhorn"]:
        mu, nu = flop_hist([tuple(e) for e in c["mu"]]), flop_hist([tuple(e) for e in c["nu"]])
        cost, it = oracle.sinkhorn_cost(mu, nu, tri)
        assert bits([cost])[0] == c["cost_bits"] and it == c["iterations"]
        assert bits([oracle.sinkhorn_divergence(mu, nu, tri)])[0] == c["divergence_bits"]
    pts = turn_like_points(8, bins=101, mass=46, seed=11).astype(np.uint32)
    for v in g["variation"]:
        assert bits([oracle.equity_variation(pts[v["i"]], pts[v["j"]])])[0] == v["bits"]


@pytest.mark.parametrize("case", load("kmeans.json"), ids=lambda c: c["kind"])
def test_oracle_reproduces_kmeans_golden(case):
    pts, tri = _points(case)
    hp = oracle.default_sinkhorn()
    hp.iterations = case["sinkhorn_iterations"]
    _run_kmeans(oracle.OracleKmeans(case["K"], pts, case["kind"], tri, hp=hp, seed=case["seed"]), case)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("case", load("mccfr_tables.json"), ids=lambda c: f"{c['game']}-{c['regret']}-{c['sampling']}")
def test_device_reproduces_mccfr_golden(gpu, case):
    from robopoker_amd.mccfr import Solver

    s = Solver(Game(case["game"]), case["regret"], case["weight"], case["sampling"], batch=case["batch"],
               seed=case["seed"], hyper=_hyper(case))
    for _ in range(case["steps"]):
        s.step()
    _check_tables(s.export(), case)
    assert list(s.counters()) == case["counters"]
    assert bits([s.exploitability()])[0] == case["exploitability_bits"]


@pytest.mark.gpu
def test_device_reproduces_sinkhorn_golden(gpu):
    from robopoker_amd import lloyd

    g = load("sinkhorn.json")
    tri = flop_metric()
    mu = np.stack([flop_hist([tuple(e) for e in c["mu"]]) for c in g["sinkhorn"]])
    nu = np.stack([flop_hist([tuple(e) for e in c["nu"]]) for c in g["sinkhorn"]])
    cost, it = lloyd.sinkhorn_cost(mu, nu, tri)
    div = lloyd.sinkhorn_divergence(mu, nu, tri)
    assert bits(cost) == [c["cost_bits"] for c in g["sinkhorn"]]
    assert it.tolist() == [c["iterations"] for c in g["sinkhorn"]]
    assert bits(div) == [c["divergence_bits"] for c in g["sinkhorn"]]
    pts = turn_like_points(8, bins=101, mass=46, seed=11).astype(np.uint32)
    x = np.stack([pts[v["i"]] for v in g["variation"]])
    y = np.stack([pts[v["j"]] for v in g["variation"]])
    assert bits(lloyd.equity_variation(x, y)) == [v["bits"] for v in g["variation"]]


@pytest.mark.gpu
@pytest.mark.parametrize("case", load("kmeans.json"), ids=lambda c: c["kind"])
def test_device_reproduces_kmeans_golden(gpu, case):
    from robopoker_amd import lloyd

    pts, tri = _points(case)
    hp = oracle.default_sinkhorn()
    hp.iterations = case["sinkhorn_iterations"]
    _run_kmeans(lloyd.Layer(case["K"], pts, case["kind"], tri, hp=hp, seed=case["seed"]), case)


def _deuce_from_oracle():
    import oracle_deuce as od
    g = load("deuce.json")
    assert [od.strength_key(h) for h in g["hands"]] == g["strength_keys"]
    assert [od.obs_i64(*od.isomorphism(*od.obs_from_i64(o))) for o in g["obs"]] == g["canonical"]
    eq = [od.river_equity(*od.obs_from_i64(o)) for o in g["river"]]
    assert bits([e[0] for e in eq]) == g["equity_bits"]
    assert [e[1] for e in eq] == g["won"] and [e[2] for e in eq] == g["total"]
    assert [od.quantize(e[0]) for e in eq] == g["bucket"]
    for street, want in g["lists"].items():
        v = od.isomorphisms(street, *want["pockets"])
        _check_list(v, want)
    assert od.project_river(g["turn"]).tolist() == g["turn_histograms"]


def _check_list(v, want):
    assert int(v.size) == want["n"] and v[:12].tolist() == want["head"] and v[-4:].tolist() == want["tail"]
    assert int(np.bitwise_xor.reduce(v)) == want["xor"]
    assert int(v.astype(np.uint64).sum() % (1 << 61)) == want["sum_mod"]


def test_oracle_reproduces_deuce_golden():
    _deuce_from_oracle()


@pytest.mark.gpu
def test_device_reproduces_deuce_golden(gpu):
    torch = pytest.importorskip("torch")
    from robopoker_amd import deuce

    g = load("deuce.json")
    assert deuce.hand_strength(g["hands"]).tolist() == g["strength_keys"]
    assert deuce.canonical(g["obs"]).tolist() == g["canonical"]
    river = torch.tensor(g["river"], dtype=torch.int64, device="cuda")
    e, b = deuce.river_equity(river)
    assert bits(e.cpu().numpy()) == g["equity_bits"] and b.cpu().tolist() == g["bucket"]
    for street, want in g["lists"].items():
        _check_list(deuce.isomorphisms(street, *want["pockets"]).cpu().numpy(), want)
    # the turn histograms through the full river table
    obs = deuce.isomorphisms("rive")
    _, bucket = deuce.river_equity(obs)
    table = deuce.Lookup("rive", obs, bucket)
    turn = torch.tensor(g["turn"], dtype=torch.int64, device="cuda")
    assert table.projections(turn, 101).cpu().numpy().astype(np.uint32).tolist() == g["turn_histograms"]


# ------------------------------------------------------------------------------------------------ configured shapes
# tests/golden/{mccfr_composed_big,kmeans_k256}.json (scripts/make_golden_big.py): oracle outputs at the shapes bench.py
# runs, which no per-test oracle run can afford — the composed update's two-level fold over 1024 chunks of a 2^18-tree
# batch; Elkan over Sinkhorn EMD at K = 256, bins = 256.
def _sha(a):
    import hashlib

    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _big(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated (scripts/make_golden_big.py)")
    return load(name)


def test_oracle_reproduces_big_composed_golden():
    case = _big("mccfr_composed_big.json")[1]  # the 2^17-tree case keeps the CPU suite short
    s = oracle.OracleSolver(Game(case["game"]), case["regret"], case["weight"], case["sampling"], batch=case["batch"], seed=case["seed"])
    for _ in range(case["steps"]):
        s.step_world(1)
    _check_tables(s.export(), case)
    assert list(s.counters()) == case["counters"]


@pytest.mark.gpu
@pytest.mark.parametrize("index", [0, 1])
def test_device_reproduces_big_composed_golden(gpu, index):
    from robopoker_amd.mccfr import Solver

    case = _big("mccfr_composed_big.json")[index]
    s = Solver(Game(case["game"]), case["regret"], case["weight"], case["sampling"], batch=case["batch"], seed=case["seed"])
    s.set_update_mode("composed")
    for _ in range(case["steps"]):
        s.step()
    _check_tables(s.export(), case)
    assert list(s.counters()) == case["counters"]


@pytest.mark.gpu
@pytest.mark.parametrize("refresh_bound", [False, True])
def test_device_reproduces_kmeans_k256_golden(gpu, monkeypatch, refresh_bound):
    # refresh_bound False: every stale-bound refresh is the bit-faithful solve (RP_LLOYD_NO_REFRESH_BOUND=1) and the whole Elkan state
    # — upper and lower bounds included — hashes to the committed values.  True (the default build): the interval-decided refresh
    # (csrc/refresh_bound.hpp) holds intervals in place of some upper bounds, so the bound hashes do not apply; everything the layer
    # hands out (assignments, drift, sizes, buckets, distances, centroids, rms) is still the golden's, bit for bit.
    from robopoker_amd import lloyd

    if not refresh_bound:
        monkeypatch.setenv("RP_LLOYD_NO_REFRESH_BOUND", "1")
    case = _big("kmeans_k256.json")
    pts = flop_like_points(case["N"], bins=case["bins"], mass=47, seed=case["seed"])
    km = lloyd.Layer(case["K"], pts, "sinkhorn", smooth_metric(case["bins"], 1), seed=case["seed"])
    assert km.refresh_stats()["enabled"] == (1 if refresh_bound else 0)
    km.set_centroids(np.array(case["start"], dtype=np.uint64))
    km.init_bounds()
    j, u, lo = km.bounds()
    assert j.tolist() == case["init_j"] and bits(u) == case["init_u_bits"] and _sha(lo.view(np.uint32)) == case["init_lower_sha"]
    for want in case["steps"]:
        d, sizes, moved = km.step()
        assert bits(d) == want["drift_bits"] and sizes.tolist() == want["sizes"] and moved == want["moved"]
        j, u, lo = km.bounds()
        assert _sha(j) == want["j_sha"]
        if not refresh_bound:
            assert _sha(u.view(np.uint32)) == want["u_sha"] and _sha(lo.view(np.uint32)) == want["lower_sha"]
    b, dist = km.lookup()
    assert b.tolist() == case["buckets"] and bits(dist) == case["distance_bits"]
    c, w = km.centroids()
    assert w.tolist() == case["centroid_weight"] and _sha(c.astype(np.uint32)) == case["centroid_sha"]
    assert bits([km.rms()])[0] == case["rms_bits"]
    st = km.prune_stats()
    assert st["enabled"] == 1 and st["survivors"] < 2 * st["points"]
    if refresh_bound:
        assert km.refresh_stats()["settled"] > 0


# ---- the NLHE blueprint traversal (tests/golden/nlmc.json, scripts/make_golden_nlmc.py): integer state and policy bits ----
def _nlmc_summary(eng, M):
    import zlib

    b = eng.batch()
    n = b["n"]
    past, present, choices, _ = eng.export()
    out = dict(n=int(n), tree=b["tree"][:n].astype(np.uint32).tolist(), n_actions=b["n_actions"][:n].astype(np.uint32).tolist(),
               expanded=b["expanded"][:n].astype(np.uint32).tolist(),
               policy_crc=zlib.crc32(np.ascontiguousarray(b["policy"][:n]).view(np.uint32).tobytes()), keys_after_batch=int(len(past)),
               keys_after_batch_crc=zlib.crc32(np.array(sorted(zip(past.tolist(), present.tolist(), choices.tolist())), dtype=np.uint64).tobytes()))
    return out


def _nlmc_table(eng):
    import zlib

    past, present, choices, enc = eng.export()
    order = np.lexsort((choices, present, past))
    return dict(table_rows=int(len(past)),
                table_keys_crc=zlib.crc32(np.stack([past[order], present[order].astype(np.uint64), choices[order]], axis=1).tobytes()),
                table_visits_crc=zlib.crc32(np.ascontiguousarray(enc[order]["visits"]).tobytes()))


NLMC = json.load(open(os.path.join(GOLD, "nlmc.json")))["cases"] if os.path.exists(os.path.join(GOLD, "nlmc.json")) else []


def _nlmc_hyper(case):
    import oracle

    hp = oracle.default_hyper()
    for k, v in case.get("hyper", {}).items():
        setattr(hp, k, v)
    return hp


@pytest.mark.parametrize("case", NLMC, ids=lambda c: f"b{c['batch']}s{c['seed']}{c.get('sampling', '')}")
def test_oracle_reproduces_nlmc_golden(case):
    import oracle_nlmc as M

    o = M.OracleNlhe(cap_log2=16, batch=case["batch"], seed=case["seed"], sampling=case.get("sampling", "external"), hyper=_nlmc_hyper(case))
    for k, v in _nlmc_summary(o, M).items():
        assert case[k] == v, k
    for _ in range(2):
        o.step()
    assert list(o.counters()) == case["counters"] and o.epoch == case["epoch"]
    for k, v in _nlmc_table(o).items():
        assert case[k] == v, k


@pytest.mark.gpu
@pytest.mark.parametrize("case", NLMC[:1] + NLMC[2:], ids=lambda c: f"b{c['batch']}s{c['seed']}{c.get('sampling', '')}")
def test_device_reproduces_nlmc_golden(gpu, case):
    # the first batch of the device equals the fixture in every integer and in the policy bits (the regret vectors carry the
    # stated tolerance and are not in the fixture; the steps after it depend on them through sampling, see test_gpu_nlmc.py)
    import oracle_nlmc as M

    from robopoker_amd.nlhe import NlheSolver

    d = NlheSolver(cap_log2=16, batch=case["batch"], seed=case["seed"], sampling=case.get("sampling", "external"), hyper=_nlmc_hyper(case))
    for k, v in _nlmc_summary(d, M).items():
        assert case[k] == v, k
    d.close()
