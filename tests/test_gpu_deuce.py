"""MI355X parity of the abstraction inputs (robopoker_amd/csrc/deuce.hip) against the CPU oracle and the reference's
published counts.  Integer work: every comparison is exact; the one float (river equity) is compared bit for bit."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import oracle_deuce as od  # noqa: E402
from test_oracle_deuce import EVALUATOR_KATS, ISOMORPHISM_KATS  # noqa: E402


@pytest.fixture(scope="module")
def deuce():
    from robopoker_amd import deuce as d
    return d


def _random_obs(rng, n_board):
    cards = rng.sample(range(52), 2 + n_board)
    return sum(1 << c for c in cards[:2]), sum(1 << c for c in cards[2:])


def test_strength_matches_oracle(deuce):
    rng = random.Random(1)
    hands = [od.hand(k[0]) for k in EVALUATOR_KATS]
    for size in (5, 6, 7):
        hands += [sum(1 << c for c in rng.sample(range(52), size)) for _ in range(60000)]
    # hands dense in one suit / few ranks: flushes, straight flushes, quads and full houses are rare at random
    for _ in range(20000):
        suit = rng.randrange(4)
        ranks = rng.sample(range(13), rng.randint(4, 7))
        h = sum(1 << (r * 4 + suit) for r in ranks)
        while bin(h).count("1") < 7:
            h |= 1 << rng.randrange(52)
        if bin(h).count("1") == 7:
            hands.append(h)
    for _ in range(20000):
        ranks = rng.sample(range(13), 3)
        pool = [r * 4 + s for r in ranks for s in range(4)]
        hands.append(sum(1 << c for c in rng.sample(pool, 7)))
    got = deuce.hand_strength(hands)
    want = np.array([od.strength_key(h) for h in hands], dtype=np.uint32)
    assert np.array_equal(got, want)
    assert len(set((want >> 21).tolist())) == 9  # every ranking variant occurred


def test_canonical_matches_oracle(deuce):
    rng = random.Random(2)
    obs = [od.obs_i64(*od.obs(a)) for pair in ISOMORPHISM_KATS for a in pair]
    for n_board in (0, 3, 4, 5):
        obs += [od.obs_i64(*_random_obs(rng, n_board)) for _ in range(20000)]
    got = deuce.canonical(obs)
    want = np.array([od.obs_i64(*od.isomorphism(*od.obs_from_i64(o))) for o in obs], dtype=np.int64)
    assert np.array_equal(got, want)
    assert np.array_equal(deuce.canonical(got), got)  # idempotent


def test_isomorphism_counts_are_the_references(deuce):
    # street.rs:120-127
    for street, n in zip(("pref", "flop", "turn", "rive"), deuce.N_ISOMORPHISMS):
        assert deuce.count_isomorphisms(street) == n


def test_isomorphism_lists_match_oracle(deuce):
    assert np.array_equal(deuce.isomorphisms("pref").cpu().numpy(), od.isomorphisms("pref"))
    assert np.array_equal(deuce.isomorphisms("flop").cpu().numpy(), od.isomorphisms("flop"))
    assert np.array_equal(deuce.isomorphisms("turn", 100, 104).cpu().numpy(), od.isomorphisms("turn", 100, 104))
    assert np.array_equal(deuce.isomorphisms("turn", 1300, 1326).cpu().numpy(), od.isomorphisms("turn", 1300, 1326))
    assert np.array_equal(deuce.isomorphisms("rive", 700, 701).cpu().numpy(), od.isomorphisms("rive", 700, 701))
    assert deuce.isomorphisms("rive", 5, 5).numel() == 0


def test_isomorphism_shards_concatenate(deuce):
    whole = deuce.isomorphisms("turn")
    assert whole.numel() == 13_960_050
    parts = [deuce.isomorphisms("turn", *deuce.shard_pockets(r, 8)) for r in range(8)]
    assert torch.equal(torch.cat(parts), whole)
    assert bool((deuce.canonical(whole[::5003].cpu().numpy()) == whole[::5003].cpu().numpy()).all())


def test_river_equity_matches_oracle(deuce):
    rng = random.Random(3)
    cases = [od.obs("2c 2d~Ts Js Qs Ks As"), od.obs("Ah Ad~As Ac Kd 7h 2c"), od.obs("2c 3d~5h 7s 9c Jd Kh")]
    cases += [_random_obs(rng, 5) for _ in range(3000)]
    obs = torch.tensor([od.obs_i64(*c) for c in cases], dtype=torch.int64, device="cuda")
    e, b = deuce.river_equity(obs)
    want = [od.river_equity(*c)[0] for c in cases]
    assert np.array_equal(e.cpu().numpy().view(np.uint32), np.array(want, dtype=np.float32).view(np.uint32))
    assert np.array_equal(b.cpu().numpy(), np.array([od.quantize(w) for w in want], dtype=np.uint8))
    assert e[0].item() == 0.5 and e[1].item() == 1.0


def test_river_equity_rejects_other_streets(deuce):
    from robopoker_amd._lib import RpError
    obs = torch.tensor([od.obs_i64(*_random_obs(random.Random(4), 4))], dtype=torch.int64, device="cuda")
    with pytest.raises(RpError):
        deuce.river_equity(obs)


@pytest.fixture(scope="module")
def river(deuce):
    """Lookup::grow(Street::Rive): all 123 156 254 river isomorphisms with their quantised equity."""
    obs = deuce.isomorphisms("rive")
    e, b = deuce.river_equity(obs)
    return obs, e, b, deuce.Lookup("rive", obs, b)


def test_river_table_properties(deuce, river):
    obs, e, b, _ = river
    assert obs.numel() == 123_156_254
    assert float(e.min()) >= 0.0 and float(e.max()) <= 1.0 and int(b.max()) == 100
    # quantise(equity) is the bucket everywhere (abstraction.rs:61-63)
    x = e * 100.0
    fl = torch.floor(x)
    assert torch.equal(torch.where(x - fl >= 0.5, fl + 1.0, fl).to(torch.uint8), b)  # f32::round: half away from zero
    # a sample against the oracle
    idx = torch.arange(0, obs.numel(), 61_578, device="cuda")
    sample = obs[idx].cpu().numpy()
    want = np.array([od.river_equity(*od.obs_from_i64(int(o)))[0] for o in sample], dtype=np.float32)
    assert np.array_equal(e[idx].cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_lookup_finds_any_suit_labelling(deuce, river):
    _, _, _, table = river
    rng = random.Random(5)
    cases = [_random_obs(rng, 5) for _ in range(2000)]
    obs = torch.tensor([od.obs_i64(*c) for c in cases], dtype=torch.int64, device="cuda")
    got = table.lookup(obs).cpu().numpy()
    want = np.array([od.quantize(od.river_equity(*c)[0]) for c in cases], dtype=np.uint8)
    assert np.array_equal(got, want)


def test_turn_points_match_the_definition(deuce, river):
    """Lookup::projections for the turn layer: histograms over the 101 river buckets (lookup.rs:27-45)."""
    _, _, _, table = river
    turn = deuce.isomorphisms("turn")
    pts = table.projections(turn, deuce.RIVER_BUCKETS)
    assert pts.shape == (13_960_050, 101)
    assert bool((pts.sum(dim=1, dtype=torch.int32) == 46).all())  # Street::Turn.n_children() (street.rs:112-118)
    idx = torch.arange(0, turn.numel(), 46_533, device="cuda")
    want = od.project_river(turn[idx].cpu().numpy())
    assert np.array_equal(pts[idx].cpu().numpy().astype(np.uint32), want)


def test_flop_points_match_oracle_through_a_turn_table(deuce):
    """Lookup::projections for the flop layer, with a synthetic (hashed) turn abstraction of 200 clusters."""
    turn = deuce.isomorphisms("turn")
    abs_ = (((turn * 2654435761) >> 20) % 200).to(torch.uint8)  # any deterministic labelling
    table = deuce.Lookup("turn", turn, abs_)
    flop = deuce.isomorphisms("flop")
    pts = table.projections(flop, 200)
    assert bool((pts.sum(dim=1, dtype=torch.int32) == 47).all())  # Street::Flop.n_children()
    idx = torch.arange(0, flop.numel(), 2_573, device="cuda")
    want = od.project(flop[idx].cpu().numpy(), turn.cpu().numpy(), abs_.cpu().numpy(), 200)
    assert np.array_equal(pts[idx].cpu().numpy().astype(np.uint32), want)
    from robopoker_amd._lib import RpError
    with pytest.raises(RpError):  # bucket indices do not fit the requested histogram
        table.projections(flop[:100], 100)
    with pytest.raises(RpError):  # turn observations have no children in a turn table
        table.projections(turn[:100], 200)


def test_lookup_rejects_tables_out_of_iterator_order(deuce):
    from robopoker_amd._lib import RpError
    flop = deuce.isomorphisms("flop")
    abs_ = torch.zeros(flop.numel(), dtype=torch.uint8, device="cuda")
    deuce.Lookup("flop", flop, abs_).close()
    with pytest.raises(RpError):
        deuce.Lookup("flop", flop.flip(0), abs_)
    with pytest.raises(RpError):
        deuce.Lookup("turn", flop, abs_)


def test_turn_points_feed_the_kmeans_layer(deuce, river):
    """The projection's output is the `counts` layout of rp_kmeans_create_device: cluster a slice with the variation
    metric on the GPU and on the oracle, from the same seeds -> identical assignments."""
    import oracle
    from robopoker_amd.lloyd import Layer
    _, _, _, table = river
    turn = deuce.isomorphisms("turn")[:5000].contiguous()
    pts = table.projections(turn, deuce.RIVER_BUCKETS)
    assert pts.shape[0] > 4096
    n = 4096
    pts = pts[:n].contiguous()
    host = pts.cpu().numpy()
    g = Layer(8, None, kind="variation", seed=9, counts_dev_ptr=pts.data_ptr(), shape=tuple(pts.shape))
    o = oracle.OracleKmeans(8, host, kind="variation", seed=9)
    g.init_centroids(), o.init_centroids()
    g.init_bounds(), o.init_bounds()
    for _ in range(3):
        g.step(), o.step()
    assert np.array_equal(g.assign()[0], o.assign()[0])


def test_pretraining_layer_and_artifact_files(deuce, river, tmp_path):
    """Layer::cluster on a slice of the real turn points (robopoker_amd.pretraining), against the oracle driven through
    the same steps, then the artifacts through the reference's row format and back."""
    import oracle
    from robopoker_amd import formats, pretraining
    obs, _, bucket, _ = river
    below = pretraining.Artifacts("rive", obs, bucket)
    art = pretraining.cluster_layer("turn", below, K=12, iterations=3, seed=4, limit=6000)
    assert art.obs.numel() == 6000 and art.metric.shape == (12 * 11 // 2,) and art.future.shape == (12, 101)
    table = deuce.Lookup("rive", obs, bucket)
    pts = table.projections(art.obs, deuce.RIVER_BUCKETS).cpu().numpy()
    o = oracle.OracleKmeans(12, pts, kind="variation", seed=4)
    o.init_centroids(), o.init_bounds()
    for _ in range(3):
        o.step()
    assert np.array_equal(art.abstraction.cpu().numpy(), o.assign()[0])
    assert np.array_equal(art.metric.view(np.uint32), o.metric().view(np.uint32))
    oc, ow = o.centroids()
    assert np.array_equal(art.future, oc) and np.array_equal(art.future_weight, ow)
    files = formats.save_artifacts(str(tmp_path), art)
    o2, a2 = formats.read_rows(files["lookup"], "qh")
    assert np.array_equal(o2, art.obs.cpu().numpy())
    assert np.array_equal(a2, (2 << 8) | art.abstraction.cpu().numpy().astype(np.int16))
    t2, d2 = formats.read_rows(files["metric"], "if")
    assert np.array_equal(d2, art.metric) and np.array_equal(t2.view(np.uint32), (2 << 30) | np.arange(66, dtype=np.uint32))
    prev, nxt, dx = formats.read_rows(files["transitions"], "hhf")
    assert set(prev.tolist()) <= {2 << 8 | k for k in range(12)} and (nxt >> 8 == 3).all()
    for k in range(12):  # each centroid's transition probabilities sum to one and come sorted
        mine = dx[prev == (2 << 8 | k)]
        if len(mine):
            assert abs(float(mine.sum()) - 1.0) < 1e-4 and (np.diff(mine) <= 0).all()


def test_sharded_pretraining_matches_single_process(deuce, river):
    """robopoker_amd.pretraining's one-process-per-GPU path (RCCL group of one rank here; the 8-GPU launch is the same
    code with eight slices): artifacts equal to the single-process layer, bit for bit."""
    import torch.distributed as dist
    from robopoker_amd import pretraining
    obs, _, bucket, _ = river
    below = pretraining.Artifacts("rive", obs, bucket)
    single = pretraining.cluster_layer("turn", below, K=10, iterations=3, seed=6, limit=5000)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
    try:
        torch.cuda.set_device(0)
        sharded = pretraining.cluster_layer_sharded("turn", below, K=10, iterations=3, seed=6, limit=5000)
        riv = pretraining.cluster_river_sharded(0)
    finally:
        if created:
            dist.destroy_process_group()
    assert torch.equal(sharded.abstraction, single.abstraction)
    assert np.array_equal(sharded.metric.view(np.uint32), single.metric.view(np.uint32))
    assert np.array_equal(sharded.future, single.future) and np.array_equal(sharded.future_weight, single.future_weight)
    assert torch.equal(riv.abstraction, bucket)
    # slices of a list tile it exactly, whatever the world size
    for world in (2, 3, 8):
        cuts = [pretraining._slice(13_960_050, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == 13_960_050
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:])) and len({c[2] for c in cuts}) == 1


def test_edge_cases_of_the_abstraction_inputs(deuce, river):
    import ctypes as C
    from robopoker_amd import _lib
    from robopoker_amd._lib import RpError
    lib = _lib.load()
    # empty inputs are fine and touch nothing
    assert deuce.hand_strength([]).size == 0 and deuce.canonical([]).size == 0
    e, b = deuce.river_equity(torch.empty(0, dtype=torch.int64, device="cuda"))
    assert e.numel() == 0 and b.numel() == 0
    # a short output buffer gets the first `cap` isomorphisms and the full count
    whole = deuce.isomorphisms("flop")
    part = torch.full((1000,), -1, dtype=torch.int64, device="cuda")
    n = C.c_uint64()
    torch.cuda.synchronize()
    _lib.check(lib.rp_isomorphisms(0, 1, 0, 1326, part.data_ptr(), 600, C.byref(n)))
    assert n.value == 1_286_792 and torch.equal(part[:600], whole[:600]) and bool((part[600:] == -1).all())
    # out-of-range street, inverted pocket range
    assert lib.rp_isomorphisms(0, 7, 0, 1326, None, 0, C.byref(n)) != 0
    _lib.check(lib.rp_isomorphisms(0, 2, 900, 100, None, 0, C.byref(n)))
    assert n.value == 0
    # a lookup miss is an error, as the reference's expect() (lookup.rs:24): a flop observation in the river table
    _, _, _, table = river
    with pytest.raises(RpError):
        table.lookup(whole[:4].contiguous())
    # strength keys of every 5-card hand: 7462 distinct classes in standard poker; the reference's flush (top card only)
    # and straight-flush conventions merge some, so just pin the count this evaluator yields against the oracle's
    rng = random.Random(9)
    sample = [sum(1 << c for c in rng.sample(range(52), 5)) for _ in range(50000)]
    keys = deuce.hand_strength(sample)
    assert len(set(keys.tolist())) == len({od.strength_key(h) for h in sample})


def test_preflop_layer_points_and_metric(deuce):
    """PrefLayer: the 169 preflop histograms (19 600 flops per pocket through a flop table) and their normalised pairwise
    metric, against the oracle's projection and Sinkhorn divergence; stand-in flop labels and metric."""
    import oracle
    from lloyd_fixtures import smooth_metric
    from robopoker_amd import pretraining
    flop = deuce.isomorphisms("flop")
    labels = (((flop * 2654435761) >> 20) % 24).to(torch.uint8)
    art = pretraining.cluster_preflop(0, pretraining.Artifacts("flop", flop, labels), smooth_metric(24, 3))
    assert art.future.shape == (169, 24) and (art.future.sum(axis=1) == 19_600).all()  # Street::Pref.n_children()
    pref = art.obs.cpu().numpy()
    idx = [0, 57, 168]
    want = od.project(pref[idx], flop.cpu().numpy(), labels.cpu().numpy(), 24)
    assert np.array_equal(art.future[idx], want)
    tri = smooth_metric(24, 3)
    d = {}
    for i, j in ((5, 2), (100, 7), (168, 167)):
        a, b = art.future[i], art.future[j]
        d[(i, j)] = (np.float32(oracle.sinkhorn_divergence(a, b, tri)) + np.float32(oracle.sinkhorn_divergence(b, a, tri))) / np.float32(2)
    raw_max = max(d.values())
    assert art.metric.max() == np.float32(1.0) and art.metric.min() >= 0
    # entries keep their ratios under the common normalisation (the maximum itself is some other pair)
    t = lambda i, j: art.metric[i * (i - 1) // 2 + j]  # noqa: E731
    (i0, j0), (i1, j1) = (5, 2), (100, 7)
    assert abs(t(i0, j0) / t(i1, j1) - d[(i0, j0)] / d[(i1, j1)]) < 1e-5 and raw_max > 0
