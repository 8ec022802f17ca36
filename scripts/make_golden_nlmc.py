#!/usr/bin/env python3
"""tests/golden/nlmc.json: the NLHE MCCFR oracle (oracle/rp_oracle_nlmc.c) frozen on two seeded cases — the first batch's
Decisions (tree ids, infoset keys, action counts, expanded masks, policy bits) and the table after two Solver::steps as sorted
(key, visits) with a checksum of the regret / weight / payoff bits.  Integer state only plus policy bits: what the device must
reproduce exactly (tests/test_golden.py); the regret vectors carry the stated tolerance and stay out of the fixture."""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_nlmc as M  # noqa: E402


def case(batch, seed, sampling="external", hyper=None):
    import oracle

    hp = oracle.default_hyper()
    for k, v in (hyper or {}).items():
        setattr(hp, k, v)
    o = M.OracleNlhe(cap_log2=16, batch=batch, seed=seed, sampling=sampling, hyper=hp)
    b = o.batch()
    n = b["n"]
    past, present, choices, _ = o.export()
    out = dict(batch=batch, seed=seed, sampling=sampling, hyper=hyper or {}, n=int(n), tree=b["tree"][:n].astype(np.uint32).tolist(),
               n_actions=b["n_actions"][:n].astype(np.uint32).tolist(), expanded=b["expanded"][:n].astype(np.uint32).tolist(),
               policy_crc=zlib.crc32(np.ascontiguousarray(b["policy"][:n]).view(np.uint32).tobytes()),
               keys_after_batch=int(len(past)),
               keys_after_batch_crc=zlib.crc32(np.array(sorted(zip(past.tolist(), present.tolist(), choices.tolist())), dtype=np.uint64).tobytes()))
    for _ in range(2):
        o.step()
    past, present, choices, enc = o.export()
    order = np.lexsort((choices, present, past))
    out["counters"] = list(o.counters())
    out["epoch"] = o.epoch
    out["table_keys_crc"] = zlib.crc32(np.stack([past[order], present[order].astype(np.uint64), choices[order]], axis=1).tobytes())
    out["table_visits_crc"] = zlib.crc32(np.ascontiguousarray(enc[order]["visits"]).tobytes())
    out["table_rows"] = int(len(past))
    return out


if __name__ == "__main__":
    # the pruned schemes bite on a FRESH table when the threshold sits above the warm-start bias of raises (10) and shoves (0):
    # PrunableSampling drops them everywhere, PluribusSampling keeps those whose child is terminal and explores 30 % of the
    # (infoset, tree) pairs — masks, tree shapes and keys from the very first batch (sample/pruning.rs:44-66, pluribus.rs:72-101)
    bite = {"prune_warmup": 0, "prune_threshold": 20.0, "prune_explore": 0.3}
    doc = {"note": __doc__.split("\n")[0],
           "cases": [case(64, 5), case(150, 12), case(96, 7, "prunable", bite), case(96, 7, "pluribus", bite)]}
    path = os.path.join(ROOT, "tests", "golden", "nlmc.json")
    json.dump(doc, open(path, "w"))
    print(path, os.path.getsize(path), "bytes")
