"""The NLHE MCCFR oracle (oracle/rp_oracle_nlmc.c): structural properties of the Decisions it produces and of the table it
trains.  (The rules underneath are pinned to the reference's unit tests in test_oracle_nlhe.py; the generic MCCFR loop to
the reference's Kuhn / Leduc thresholds in test_oracle_mccfr.py — this file checks the NLHE instance of both.)"""
import ctypes as C

import numpy as np

import oracle_nlhe as rules
import oracle_nlmc as M


def _edges(path):
    out = []
    while path and (path & 0x1f):
        out.append(path & 0x1f)
        path >>= 5
    return out


def test_first_batch_decisions_are_well_formed():
    s = M.OracleNlhe(cap_log2=16, batch=48, seed=11)
    b = s.batch()
    assert b["n"] > 48  # a few dozen walker infosets per tree
    past, present, choices, enc = s.export()
    keyed = M.as_map(past, present, choices, enc)
    assert len(keyed) == s.counters()[2]
    for i in range(b["n"]):
        n = int(b["n_actions"][i])
        assert 2 <= n <= M.A and b["expanded"][i] == (1 << n) - 1  # external sampling expands every walker edge
        pol, reg = b["policy"][i, :n], b["regret"][i, :n]
        assert abs(float(pol.sum()) - 1.0) < 1e-5 and np.all(pol > 0)
        assert np.all(b["policy"][i, n:] == 0) and np.all(b["regret"][i, n:] == 0)
        # regret_a = cfv_a - ev with ev = sum sigma_a cfv_a per root: the policy-weighted regret vanishes (summed over a span too)
        assert abs(float((pol * reg).sum())) < 2e-3 * max(1.0, float(np.abs(reg).max()))
        assert abs(float(b["payoff"][i])) <= 400.0 * 64  # |payoff| <= the stacks times the span
    # trees are emitted in tree order
    assert np.all(np.diff(b["tree"].astype(np.int64)) >= 0) and b["tree"].max() < 48


def test_infoset_keys_follow_the_reference_layout():
    # NlheInfo (nlhe/src/info.rs:145-160): past = the choice edges of the current street (never a Draw), choices = the edges
    # offered there, in legal() order (raises, shove, call, fold | check); the bucket is below the street's bucket count
    s = M.OracleNlhe(cap_log2=16, batch=64, seed=3)
    s.step()
    past, present, choices, enc = s.export()
    assert len(past) > 500
    for p, b, c in zip(past, present, choices):
        pe, ce = _edges(int(p)), _edges(int(c))
        assert 1 not in pe and 1 not in ce and 2 <= len(ce) <= M.A and (b >> 8) <= 3 and (b & 0xff) < (169, 256, 256, 101)[b >> 8]
        assert len(set(ce)) == len(ce)
        raises = [e for e in ce if e >= 6]
        assert ce[: len(raises)] == raises  # the raise grid comes first (game.rs:253-283)
        assert not (2 in ce and 3 in ce)  # fold and check exclude each other
    # a fresh infoset starts at the edge-wise bias (kicker/src/edge.rs:61-72, bias.rs:47-70)
    fresh = enc["visits"].sum(axis=1) == 0
    assert fresh.any()
    bias = {2: 100.0, 3: 50.0, 4: 50.0, 5: 0.0}
    for c, row in zip(choices[fresh], enc[fresh]):
        for a, e in enumerate(_edges(int(c))):
            assert row["regret"][a] == bias.get(e, 10.0) and row["weight"][a] == 0.0


def test_steps_are_deterministic_and_counters_add_up():
    a = M.OracleNlhe(cap_log2=17, batch=32, seed=7)
    b = M.OracleNlhe(cap_log2=17, batch=32, seed=7)
    for _ in range(3):
        a.step()
        b.step()
    ea, eb = M.as_map(*a.export()), M.as_map(*b.export())
    assert ea.keys() == eb.keys() and a.counters() == b.counters() and a.epoch == 3
    for k in ea:
        assert ea[k].tobytes() == eb[k].tobytes()
    nodes, infos, keys = a.counters()
    visits = sum(int(v["visits"][0]) for v in ea.values())
    assert visits == infos  # every Decisions touches its infoset once (solver.rs:187-192)
    assert nodes > infos > 0 and keys >= len([1 for v in ea.values() if v["visits"][0]])


def test_hash_encoder_uses_the_canonical_observation():
    # suit-isomorphic observations must share a bucket: (As Kh | 2c 7d Jh) and its image under a suit permutation
    o = M.lib()
    import oracle_deuce as od

    def bucket(street, pocket, board):
        cp, cb = od.isomorphism(pocket, board)
        return o.ora_nlmc_hash_bucket(street, od.obs_i64(cp, cb))

    def card(rank, suit):
        return 1 << (4 * rank + suit)

    p1, b1 = card(12, 3) | card(11, 2), card(0, 0) | card(5, 1) | card(9, 2)
    p2, b2 = card(12, 0) | card(11, 1), card(0, 3) | card(5, 2) | card(9, 1)  # suits 3->0, 2->1, 0->3, 1->2
    assert bucket(1, p1, b1) == bucket(1, p2, b2)
    assert rules is not None


def _pruning_hyper(warmup=2, threshold=-5.0, explore=0.05):
    import oracle

    hp = oracle.default_hyper()
    hp.prune_warmup, hp.prune_threshold, hp.prune_explore = warmup, threshold, explore
    return hp


def test_pluribus_sampling_prunes_walker_edges_after_the_warm_up():
    # PluribusSampling (sample/pluribus.rs:72-101) on the NLHE path (nlhe/src/lib.rs:86-90 Flagship): no pruning during the
    # warm-up, afterwards walker edges whose raw regret is at or below the threshold are dropped unless their child is terminal
    # (a fold always is), except in the explored fraction of (infoset, tree) pairs; nothing kept = everything kept
    s = M.OracleNlhe(cap_log2=18, batch=96, seed=5, sampling="pluribus", hyper=_pruning_hyper())
    e = M.OracleNlhe(cap_log2=18, batch=96, seed=5, sampling="external", hyper=_pruning_hyper())
    for step in range(2):  # warm-up: identical to external sampling
        bs, be = s.batch(), e.batch()
        assert bs["n"] == be["n"] and np.array_equal(bs["expanded"], be["expanded"])
        assert np.array_equal(bs["regret"].view(np.uint32), be["regret"].view(np.uint32))
        s.step()
        e.step()
    assert s.counters() == e.counters()
    pruned_any = False
    for step in range(4):
        b = s.batch()
        past, present, choices, enc = s.export()
        keyed = M.as_map(past, present, choices, enc)
        full = (1 << b["n_actions"].astype(np.uint32)) - 1
        partial = b["expanded"] != full
        pruned_any |= bool(partial.any())
        assert np.all(b["expanded"] != 0) and np.all((b["expanded"] & ~full) == 0)
        # a dropped edge has raw regret <= threshold in the table and is never a fold (its child would be terminal)
        o = M.lib()
        for i in np.nonzero(partial)[0][:200]:
            n = int(b["n_actions"][i])
            kp, kb, kc = C.c_uint64(), C.c_uint32(), C.c_uint64()
            assert o.ora_nlmc_row_key(s._h, int(b["row"][i]), C.byref(kp), C.byref(kb), C.byref(kc)) == 0
            row, edges = keyed[(kp.value, kb.value, kc.value)], _edges(kc.value)
            assert np.all(b["regret"][i, n:] == 0)
            for a in range(n):
                if not (int(b["expanded"][i]) >> a) & 1:
                    assert row["regret"][a] <= -5.0 and edges[a] != 2
                    assert b["regret"][i, a] == 0.0  # an unexpanded edge receives no regret (solver.rs:263-305)
        s.step()
    assert pruned_any
    nodes_p, _, _ = s.counters()
    for _ in range(4):
        e.step()
    nodes_e, _, _ = e.counters()
    assert nodes_p < nodes_e  # pruning shrinks the trees


def test_prunable_sampling_has_no_warm_up_and_no_terminal_exemption():
    hp = _pruning_hyper(warmup=0, threshold=20.0)  # the bias leaves raises at 10 and shoves at 0: pruned from the first epoch
    s = M.OracleNlhe(cap_log2=18, batch=32, seed=9, sampling="prunable", hyper=hp)
    b = s.batch()
    full = (1 << b["n_actions"].astype(np.uint32)) - 1
    assert (b["expanded"] != full).any() and np.all(b["expanded"] != 0)
    past, present, choices, enc = s.export()
    # every infoset of a fresh table keeps exactly the edges whose bias exceeds 20: fold (100), check / call (50)
    for i in range(int(b["n"])):
        n = int(b["n_actions"][i])
        assert bin(int(b["expanded"][i])).count("1") <= n
    assert s.counters()[2] == len(past)


def test_prunable_masks_follow_from_the_table():
    # PrunableSampling (sample/pruning.rs:44-66) has no warm-up, no explore draw and no terminal exemption: which edges a
    # Decisions expanded is a pure function of its infoset's accumulated regrets — recomputed here from the exported table
    hp = _pruning_hyper(warmup=0, threshold=-2.0)
    s = M.OracleNlhe(cap_log2=18, batch=64, seed=13, sampling="prunable", hyper=hp)
    o = M.lib()
    partial = 0
    for step in range(5):
        b = s.batch()
        keyed = M.as_map(*s.export())
        for i in range(int(b["n"])):
            kp, kb, kc = C.c_uint64(), C.c_uint32(), C.c_uint64()
            assert o.ora_nlmc_row_key(s._h, int(b["row"][i]), C.byref(kp), C.byref(kb), C.byref(kc)) == 0
            row, n = keyed[(kp.value, kb.value, kc.value)], int(b["n_actions"][i])
            keep = sum(1 << a for a in range(n) if row["regret"][a] > np.float32(-2.0))
            want = keep or (1 << n) - 1
            assert int(b["expanded"][i]) == want, (step, i)
            partial += want != (1 << n) - 1
        s.step()
    assert partial > 20
