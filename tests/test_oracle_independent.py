"""The C oracle against INDEPENDENT restatements (tests/ref_numpy.py: float64 numpy written from the Rust sources) and
against closed forms — a second pin next to the reference's own threshold tests.  The oracle and the HIP kernels share
one author and one include/rp_math.h; these checks do not."""
import itertools
from fractions import Fraction

import numpy as np
import pytest

import oracle
import ref_numpy as R
from lloyd_fixtures import flop_hist, flop_like_points, flop_metric, random_metric, smooth_metric, turn_like_points
from robopoker_amd import Game


# ---------------------------------------------------------------------------------------------- Sinkhorn
def _pairs(bins, n, seed, mass=30):
    pts = flop_like_points(2 * n, bins=bins, mass=mass, seed=seed).astype(np.uint32)
    return pts[:n], pts[n:]


@pytest.mark.parametrize("bins,metric", [(32, "closed"), (64, "smooth"), (48, "random")])
def test_sinkhorn_cost_and_iterations_match_the_float64_restatement(bins, metric):
    tri = {"closed": flop_metric, "smooth": lambda b: smooth_metric(b, 3),
           "random": lambda b: random_metric(b, np.random.default_rng(7))}[metric](bins)
    C = R.dense_cost(tri, bins)
    mu, nu = _pairs(bins, 12, seed=bins)
    for a, b in zip(mu, nu):
        got, it, errs, costs = oracle.sinkhorn_trace(a, b, tri, bins=bins)
        want, wit, werrs, wcosts = R.sinkhorn_f64(a, b, C, trace=True)
        # the cost of EVERY iterate agrees to f32 accuracy (the trajectories are the same contraction)
        assert np.max(np.abs(costs - wcosts)) <= 1e-5 * max(want, 1e-2)
        # the returned cost: the stopping iteration may differ by rounding of the statistic, the value may not
        assert abs(got - want) <= 1e-5 * max(want, 1e-2) + abs(wcosts[it - 1] - want)
        # where the f32 stopping statistic resolves the true one (small potentials: no stall of the log-domain
        # iterate) both stop within an iteration of each other; elsewhere only the VALUE is comparable (above)
        k = wit - 1
        if wit < 128 and abs(errs[k] - werrs[k]) < 0.05 * werrs[k] and abs(errs[k - 1] - werrs[k - 1]) < 0.05 * werrs[k - 1]:
            assert abs(it - wit) <= 1


def test_sinkhorn_divergence_matches_float64_on_the_reference_fixture():
    # crates/lloyd/src/sinkhorn.rs:240-293
    tri = flop_metric()
    C = R.dense_cost(tri, 32)
    mu = flop_hist([(0, 3), (5, 1), (12, 4)])
    nu = flop_hist([(2, 2), (8, 5), (20, 1), (24, 3)])
    for a, b in ((mu, nu), (nu, mu), (mu, mu)):
        assert abs(oracle.sinkhorn_divergence(a, b, tri) - R.sinkhorn_divergence_f64(a, b, C)) < 1e-5


def test_sinkhorn_closed_form_single_bin_source():
    # mu = one bin x0: every coupling with the right marginals is pi(x0, y) = nu(y), so the cost is sum_y nu(y) C(x0, y)
    # whatever the regularisation (the first rhs update already has the exact column marginals)
    bins = 40
    tri = smooth_metric(bins, 9)
    C = R.dense_cost(tri, bins)
    rng = np.random.default_rng(0)
    for _ in range(10):
        mu = np.zeros(bins, dtype=np.uint32)
        x0 = int(rng.integers(bins))
        mu[x0] = int(rng.integers(1, 9))
        nu = np.zeros(bins, dtype=np.uint32)
        sup = rng.choice(bins, size=int(rng.integers(1, 20)), replace=False)
        nu[sup] = rng.integers(1, 9, size=sup.size)
        exact = float((nu / nu.sum()) @ C[x0])
        got, it = oracle.sinkhorn_cost(mu, nu, tri, bins=bins)
        assert abs(got - exact) < 2e-6
        got_t, _ = oracle.sinkhorn_cost(nu, mu, tri, bins=bins)  # and transposed
        assert abs(got_t - exact) < 2e-6


# ---------------------------------------------------------------------------------------------- variation
def test_variation_matches_rational_arithmetic():
    pts = turn_like_points(40, bins=101, mass=46, seed=2).astype(np.uint32)
    for x, y in zip(pts[:20], pts[20:]):
        exact = R.variation_exact(x, y)
        got = oracle.equity_variation(x, y)
        assert abs(got - float(exact)) < 2e-6
        assert abs(R.variation_f64(x, y) - float(exact)) < 1e-12
    # closed form: two point masses k buckets apart differ in exactly k CDF cells
    x = np.zeros(101, dtype=np.uint32)
    y = np.zeros(101, dtype=np.uint32)
    x[10], y[35] = 7, 3
    assert R.variation_exact(x, y) == Fraction(25, 101)
    assert abs(oracle.equity_variation(x, y) - 25 / 101) < 1e-6


# ---------------------------------------------------------------------------------------------- MCCFR
def test_distributions_match_the_float64_restatement():
    g = Game("leduc")
    s = oracle.OracleSolver(g, "linear", "linear", "external", batch=64, seed=3)
    s.solve(64 * 40)
    rows = s.export().reshape(g.n_infos, g.max_actions)
    for info in range(0, g.n_infos, 7):
        n = g.n_actions(info)
        assert np.allclose(s.policy(info, "iterated"), R.regret_matching_f64(rows["regret"][info, :n]), rtol=2e-6, atol=1e-7)
        assert np.allclose(s.policy(info, "averaged"), R.averaged_f64(rows["weight"][info, :n]), rtol=2e-6, atol=1e-7)
        assert np.allclose(s.policy(info, "sampling"), R.sampling_f64(rows["weight"][info, :n]), rtol=2e-6, atol=1e-7)


def _sampled_trees(g, walker):
    """every externally-sampled tree of the game for `walker`: chance and opponent states keep ONE child (all
    alternatives enumerated), walker states keep all (sample/external.rs:17-64).  A tree is a dict state -> kept children."""
    t = g.table

    def expand(state):
        st = t.states[state]
        if st.n_children == 0:
            yield {}
            return
        kids = [t.children[st.offset + a] for a in range(st.n_children)]
        if st.turn == walker:
            for combo in itertools.product(*[list(expand(k)) for k in kids]):
                tree = {state: list(enumerate(kids))}
                for sub in combo:
                    tree.update(sub)
                yield tree
        else:
            for a, k in enumerate(kids):
                for sub in expand(k):
                    tree = {state: [(a, k)]}
                    tree.update(sub)
                    yield tree

    return expand(t.train_root)


def _hand_decisions(g, tree, walker):
    """CfrFlow::dfs (flow.rs:64-87) for every walker infoset of one sampled tree at epoch 0 of a fresh profile: regret
    matching is uniform and the sampling distribution is uniform, so an opponent edge carries sigma / q = 1 and a walker
    edge sigma = 1/|choices| below the root (flow.rs:182-216); chance edges carry 1."""
    t = g.table

    def value(state):  # recursed_value with relative / sampling reach folded in (exact rationals)
        st = t.states[state]
        if st.n_children == 0:
            return Fraction(t.payoffs[st.offset * t.n_players + walker]).limit_denominator(1 << 20)
        kept = tree[state]
        if st.turn == walker:
            return sum(value(k) for _, k in kept) / st.n_children
        return sum(value(k) for _, k in kept)  # one child; sigma / q = 1 (or chance: 1)

    out = []
    for state, kept in tree.items():
        st = t.states[state]
        if st.turn != walker:
            continue
        vals = [value(k) for _, k in kept]  # ancestor_reach = 1: every non-walker ancestor edge has sigma = q
        ev = sum(vals) / len(vals)
        out.append((st.info, tuple(float(v - ev) for v in vals), float(ev)))
    return out


@pytest.mark.parametrize("game", ["kuhn", "leduc"])
def test_first_epoch_decisions_are_hand_traceable(game):
    # An independent enumeration (Python, exact rationals) of every tree external sampling can produce and of the
    # counterfactual regret vector / expected value CfrFlow::dfs assigns to each walker infoset in it, for a FRESH
    # profile (epoch 0: uniform regret matching, uniform sampling).  Every Decisions the oracle emits in its first batch
    # must be one of them — whatever its RNG drew.
    g = Game(game)
    walker = 0  # epoch 0: walker = epochs % players (book.rs)
    allowed = set()
    n_trees = 0
    for tree in _sampled_trees(g, walker):
        n_trees += 1
        for info, regrets, ev in _hand_decisions(g, tree, walker):
            allowed.add((info, tuple(round(r, 5) for r in regrets), round(ev, 5)))
        if n_trees > 60000:
            break
    s = oracle.OracleSolver(g, "summed", "constant", "external", batch=400, seed=5)
    decs = s.batch()  # Solver::batch at epoch 0: pure w.r.t. the profile (solver.rs:225-240)
    s.step()          # ... and step() applies exactly that batch
    assert len(decs) > 0
    for d in decs:
        key = (d["info"], tuple(round(r, 5) for r in d["regret"]), round(d["payoff"], 5))
        assert key in allowed, key
    # and the table after that step is the sum of those regret vectors (Summed schedule: R += delta, solver.rs:143-152)
    rows = s.export().reshape(g.n_infos, g.max_actions)
    want = np.zeros((g.n_infos, g.max_actions))
    for d in decs:
        want[d["info"], :d["n"]] += d["regret"]
    assert np.allclose(rows["regret"], want, rtol=1e-5, atol=1e-5)


# ---- the pruning schemes' masks, from the table alone (sample/pruning.rs:44-66, pluribus.rs:72-101) -----------------------
_M64 = (1 << 64) - 1


def _mix64(z):
    z ^= z >> 30
    z = (z * 0xbf58476d1ce4e5b9) & _M64
    z ^= z >> 27
    z = (z * 0x94d049bb133111eb) & _M64
    return z ^ (z >> 31)


def _node_hash(seed, epoch, tree, key):  # include/rp_math.h rp_node_hash, restated on Python integers
    h = _mix64((seed + 0x9e3779b97f4a7c15) & _M64)
    h = _mix64(h ^ ((epoch * 0xd1342543de82ef95 + 0x632be59bd9b4e019) & _M64))
    h = _mix64(h ^ ((tree * 0xaf251af3b0f025b5 + 0x2545f4914f6cdd1d) & _M64))
    # the per-draw half: Murmur3's 32-bit finaliser over the low word and the folded key, the high word multiplied in
    lo, hi, k = h & 0xffffffff, h >> 32, (key ^ (key >> 32)) & 0xffffffff
    x = lo ^ k
    x ^= x >> 16
    x = (x * 0x85ebca6b) & 0xffffffff
    x ^= x >> 13
    x = (x * 0xc2b2ae35) & 0xffffffff
    x ^= x >> 16
    x = ((x ^ hi) * 0x9e3779b1) & 0xffffffff
    x ^= x >> 15
    return x << 32


def _u01(h):  # rp_u01: the top 24 bits as a float in [0, 1)
    return np.float32(h >> 40) * np.float32(5.9604644775390625e-8)


@pytest.mark.parametrize("sampling", ["pluribus", "prunable"])
def test_pruned_masks_follow_from_the_table(sampling):
    # Which walker edges a Decisions expanded is a pure function of (the infoset's accumulated regrets, the scheme's constants,
    # the epoch, one hashed draw per (epoch, infoset, tree), which children are terminal): recomputed here from the exported table
    # and the game table, in Python, for every Decisions of six Leduc batches with pruning forced to bite.
    g = Game("leduc")
    t = g.table
    hp = oracle.default_hyper()
    hp.prune_warmup, hp.prune_threshold, hp.prune_explore = 2, -0.05, 0.3
    seed = 77
    s = oracle.OracleSolver(g, "linear", "linear", sampling, batch=300, seed=seed, hyper=hp)
    # terminal children per infoset (the same for every state of an infoset: the actions of a betting state are public)
    term = {}
    for st_i in range(t.n_states):
        st = t.states[st_i]
        if st.turn >= t.n_players:
            continue
        bits = 0
        for a in range(st.n_children):
            child = t.states[t.children[st.offset + a]]
            bits |= (1 << a) if child.turn == 255 else 0
        assert term.setdefault(st.info, bits) == bits
    pruned = explored = 0
    for step in range(6):
        rows = s.export().reshape(g.n_infos, g.max_actions)
        epoch = s.epoch
        for d in s.batch():
            n, info = d["n"], d["info"]
            full = (1 << n) - 1
            want = full
            live = sampling == "prunable" or epoch >= hp.prune_warmup
            if live and sampling == "pluribus" and _u01(_node_hash(seed, epoch, d["tree"], info)) < np.float32(hp.prune_explore):
                live = False
                explored += 1
            if live:
                keep = 0
                for a in range(n):
                    if rows["regret"][info, a] > np.float32(hp.prune_threshold) or (sampling == "pluribus" and (term[info] >> a) & 1):
                        keep |= 1 << a
                want = keep or full
            assert d["expanded"] == want, (step, d, want)
            pruned += want != full
        s.step()
    assert pruned > 20 and (sampling == "prunable" or explored > 20)
