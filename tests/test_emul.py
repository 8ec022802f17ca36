"""The kernels' logic on a machine without a GPU: a short selection of the `-m gpu` parity tests, run through the wave64
execution model of tests/emul/ (DESIGN.md §2b).  Test infrastructure checking test subjects — the product library is not
involved, and nothing here is a measurement.

The selection runs in a child pytest (RP_EMUL=1 swaps the handle of the ctypes binding for that process only): a few bit-exact
comparisons with the oracle per kernel family, chosen to finish in about a minute after the ~40 s build."""
from __future__ import annotations

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = os.environ.get("RP_EMUL_CXX", "/opt/rocm/lib/llvm/bin/clang++")

SELECTION = [
    # Path A, dense solver: ordered update, composed update, the skeleton-instantiated traversal, pruned sampling
    ("tests/test_gpu_mccfr.py", "(test_tables_bit_exact_vs_oracle and (kuhn or rps)) or (composed_mode_matches_oracle_world_semantics and 777)"
                                " or test_static_skeleton_traversal_equals_generic_and_oracle or test_hyper_parameter_corners_bit_exact"),
    # the sparse profile: own radix sort / scan / run lengths, ordered and composed application
    ("tests/test_gpu_sparse.py", "test_ordered_apply_bit_exact or test_composed_apply or (test_batches_just_past_one_scan_tile_set_bit_exact and 66000)"
                                 " or test_small_batches_prepared_by_one_workgroup_bit_exact"),  # round 6: k_prep_one
    # NLHE traversal: level-synchronous expansion against the oracle, pruned schemes, ragged batches, the chunked retry
    ("tests/test_gpu_nlmc.py", "(test_first_batch_equals_the_oracle and 64) or test_pruned_sampling_schemes_equal_the_oracle"
                               " or test_ragged_batches_equal_the_oracle or test_a_batch_traversed_in_several_passes"
                               " or test_a_full_infoset_table_fails"),
    # round 4: reference-seed mode (DefaultHasher -> SmallRng draws in the skeleton and the generic kernels, k-means++'s sequential f32
    # running sums), the one-tree-per-workgroup NLHE traversal with the values in the reference's order, the same order on the level kernels
    ("tests/test_gpu_mccfr.py", "(test_reference_seed_tables_bit_exact_vs_oracle and (kuhn or rps)) or test_reference_seed_in_the_generic_traversals"),
    ("tests/test_gpu_nlmc.py", "(test_small_batch_decisions_are_bit_exact and 33) or (test_exact_order_on_the_level_synchronous_kernels and 160)"
                               " or (test_tree_per_workgroup_traversal_equals_the_level_synchronous_one and external-1)"
                               " or (test_reference_seed_mode_equals_the_oracle and external)"),
    ("tests/test_gpu_lloyd.py", "test_reference_seed_kmeanspp_picks_equal_the_oracle and (1024 or variation-3-5)"),
    # the stand-alone Sinkhorn operators in glibc's arithmetic (expf / logf evaluated in double: include/rp_libm_glibc.h)
    ("tests/test_gpu_z_glibc_mode.py", "test_closed_form_fixture or (test_random_pairs and 32-5-9) or test_converging_solves or test_the_mode_differs"
                                       " or (test_a_layer_clustered_in_glibc_arithmetic and (5-150 or reference)) or test_the_layer_mode_is_set"
                                       " or test_glibc_expf_and_logf_on_the_device or test_the_pruned_glibc_pass or test_the_kernels_branch_free_glibc_forms"),
    # Path B: wave-cooperative Sinkhorn, Elkan iterations with remembered pairwise entries
    ("tests/test_gpu_lloyd.py", "(test_sinkhorn_random_pairs_bit_exact and 32-5-9) or test_sinkhorn_fixture_bit_exact"
                                " or (test_elkan_iterations_bit_exact and sinkhorn-5-150) or test_equity_variation_bit_exact"
                                " or test_empty_histogram_costs_zero or (test_layer_shape_corners_bit_exact and sinkhorn-65-130)"
                                " or test_interval_decided_refresh_keeps_the_reference_state"),
]


@pytest.fixture(scope="module")
def emulated_library():
    if not os.path.exists(CLANG):
        pytest.skip(f"{CLANG} (host compiler of the execution model) is not installed")
    sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
    import build as emul_build

    return emul_build.build(jobs=os.cpu_count() or 4)


CASES = [(c[0], c[1], c[2] if len(c) > 2 else {}) for c in SELECTION]


@pytest.mark.parametrize("module,expr,extra", CASES, ids=[c[0].split("/")[-1][9:-3] + ("-" + "-".join(k.split("_", 2)[-1].lower() for k in c[2]) if c[2] else "") for c in CASES])
def test_kernel_sources_under_the_wave64_model(emulated_library, module, expr, extra):
    env = dict(os.environ, RP_EMUL="1", RP_EMUL_GUARD="1", RP_EMUL_TRAP="1", **extra)
    r = subprocess.run([sys.executable, "-m", "pytest", module, "-m", "gpu", "-q", "-x", "-k", expr, "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and " failed" not in r.stdout, tail


def test_the_emulated_library_exports_the_whole_abi(emulated_library):
    import ctypes as C

    from robopoker_amd import _lib

    lib = C.CDLL(emulated_library)
    missing = [n for n in _lib.declared_symbols() if not hasattr(lib, n)]
    assert not missing


@pytest.mark.parametrize("order", ["forward", "reverse", "random"])
def test_the_execution_model_keeps_its_own_rules(emulated_library, order):
    # tests/emul/selfcheck.cpp: ballots in a loop with per-lane trip counts, a shuffle after a divergent shuffle (the lanes that
    # skipped the branch wait for the others), sources outside EXEC, segment widths, a barrier after a wavefront has left,
    # readfirstlane under a partial mask, the 16x16x4 MFMA layout against a host matmul, 500 workgroups on one counter with
    # their own LDS — under every resume order
    code = "import ctypes as C; raise SystemExit(C.CDLL(%r).emu_selfcheck())" % emulated_library
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RP_EMUL_ORDER=order), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
