"""ctypes binding of librp_mi355x.so (include/rp_mi355x.h).

This is the Python face of the C-ABI: plain pointers and sizes, no torch types.  The library is
built in-tree by ``__graft_entry__.build()`` (``make -C robopoker_amd/csrc``); importing this module
when the shared object is missing raises — there is no CPU fallback for the product path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librp_mi355x.so")

RP_OK = 0
RP_ERR_INVALID = 1
RP_ERR_NO_DEVICE = 2
RP_ERR_HIP = 3
RP_ERR_UNSUPPORTED = 4
RP_ERR_CAPACITY = 5

RP_TURN_CHANCE = 254
RP_TURN_TERMINAL = 255
RP_NO_INFO = 0xFFFFFFFF

# enums (rp_regret_kind, rp_weight_kind, rp_sampling_kind, rp_game_kind, rp_dist_kind, rp_metric_kind)
REGRET = {"summed": 0, "linear": 1, "discounted": 2, "floored": 3, "asymmetric": 4}
WEIGHT = {"constant": 0, "linear": 1, "quadratic": 2, "exponential": 3}
SAMPLING = {"external": 0, "prunable": 1, "pluribus": 2}
GAME = {"kuhn": 0, "leduc": 1, "rps": 2, "leduc_wide": 3}
DIST = {"iterated": 0, "averaged": 1, "sampling": 2}
METRIC = {"sinkhorn": 0, "variation": 1}
UPDATE = {"ordered": 0, "composed": 1}


class Hyper(C.Structure):
    _fields_ = [
        ("temperature", C.c_float),
        ("smoothing", C.c_float),
        ("curiosity", C.c_float),
        ("prune_threshold", C.c_float),
        ("prune_explore", C.c_float),
        ("prune_warmup", C.c_uint64),
        ("regret_min", C.c_float),
        ("_pad", C.c_uint32),
    ]


class Encounter(C.Structure):
    _fields_ = [("weight", C.c_float), ("regret", C.c_float), ("payoff", C.c_float), ("visits", C.c_uint32)]


class State(C.Structure):
    _fields_ = [
        ("turn", C.c_uint8),
        ("n_children", C.c_uint8),
        ("chance_info", C.c_uint16),
        ("info", C.c_uint32),
        ("offset", C.c_uint32),
    ]


class HashStream(C.Structure):
    """rp_hash_stream: what `impl Hash for I` writes for one infoset (reference-seed mode)"""
    _fields_ = [("len", C.c_uint8), ("bytes", C.c_uint8 * 55)]


class HashStreams(C.Structure):
    _fields_ = [("n_infos", C.c_uint32), ("n_chance", C.c_uint32), ("infos", C.POINTER(HashStream)),
                ("chance", C.POINTER(HashStream))]


RNG = {"counter": 0, "reference": 1}


class GameTable(C.Structure):
    _fields_ = [
        ("n_states", C.c_uint32),
        ("n_infos", C.c_uint32),
        ("n_players", C.c_uint32),
        ("max_actions", C.c_uint32),
        ("n_children", C.c_uint32),
        ("n_terminals", C.c_uint32),
        ("train_root", C.c_uint32),
        ("exploit_root", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("max_tree_nodes", C.c_uint32),
        ("states", C.POINTER(State)),
        ("children", C.POINTER(C.c_uint32)),
        ("payoffs", C.POINTER(C.c_float)),
        ("info_actions", C.POINTER(C.c_uint8)),
        ("info_player", C.POINTER(C.c_uint8)),
        ("default_regret", C.POINTER(C.c_float)),
    ]


class Decisions(C.Structure):
    """rp_decisions: one batch of Decisions in DEVICE memory (raw pointers)."""
    _fields_ = [
        ("n", C.c_uint32),
        ("row", C.c_void_p),
        ("n_actions", C.c_void_p),
        ("expanded", C.c_void_p),
        ("regret", C.c_void_p),
        ("policy", C.c_void_p),
        ("payoff", C.c_void_p),
    ]


class SinkhornHP(C.Structure):
    _fields_ = [("temperature", C.c_float), ("iterations", C.c_uint32), ("tolerance", C.c_float)]


class RpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rp_mi355x error {code}: {msg}")
        self.code = code


# every symbol include/rp_mi355x.h declares (tests/test_abi.py checks the list against the header)
class PruneStats(C.Structure):
    """rp_prune_stats (include/rp_mi355x.h): what the MFMA Sinkhorn bound discarded and what it cost"""
    _fields_ = [("enabled", C.c_uint32), ("reserved", C.c_uint32), ("points", C.c_uint64), ("candidates", C.c_uint64),
                ("survivors", C.c_uint64), ("block_iterations", C.c_uint64), ("cost_passes", C.c_uint64),
                ("mfma_instructions", C.c_uint64), ("audited_points", C.c_uint64), ("audit_mismatches", C.c_uint64),
                ("sampled_points", C.c_uint64), ("sample_mismatches", C.c_uint64), ("kpp_bound_pairs", C.c_uint64),
                ("kpp_bound_kept", C.c_uint64), ("kpp_bound_iterations", C.c_uint64), ("kpp_bound_cost_passes", C.c_uint64),
                ("column_iterations", C.c_uint64), ("ref_pick_chunks", C.c_uint64), ("ref_pick_walked", C.c_uint64)]


_SIGNATURES = {
    "rp_last_error": (C.c_char_p, []),
    "rp_device_count": (C.c_int, []),
    "rp_version": (C.c_char_p, []),
    "rp_math_selftest": (C.c_int, [C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rp_libm_glibc_sweep": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]),
    "rp_libm_glibc_tab_sweep": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]),
    "rp_math_exp_sweep": (C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
    "rp_sortscan_selftest": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32] + [C.c_void_p] * 8),
    "rp_hyper_default": (None, [C.POINTER(Hyper)]),
    "rp_game_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "rp_game_view": (C.c_int, [C.c_void_p, C.POINTER(GameTable)]),
    "rp_game_destroy": (C.c_int, [C.c_void_p]),
    "rp_game_info_id": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint32)]),
    "rp_game_info_name": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]),
    "rp_game_table_check": (C.c_int, [C.POINTER(GameTable)]),
    "rp_mccfr_create": (
        C.c_int,
        [C.POINTER(GameTable), C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(Hyper), C.c_uint64, C.c_int,
         C.POINTER(C.c_void_p)],
    ),
    "rp_mccfr_create_mode": (
        C.c_int,
        [C.POINTER(GameTable), C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(Hyper), C.c_uint64, C.c_int, C.c_int,
         C.POINTER(C.c_void_p)],
    ),
    "rp_mccfr_destroy": (C.c_int, [C.c_void_p]),
    "rp_mccfr_step": (C.c_int, [C.c_void_p]),
    "rp_mccfr_solve": (C.c_int, [C.c_void_p, C.c_uint64]),
    "rp_mccfr_spend": (C.c_int, [C.c_void_p, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "rp_mccfr_step_async": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rp_mccfr_sync": (C.c_int, [C.c_void_p]),
    "rp_mccfr_epoch": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rp_mccfr_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rp_mccfr_get": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Encounter)]),
    "rp_mccfr_set": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Encounter)]),
    "rp_mccfr_export": (C.c_int, [C.c_void_p, C.POINTER(Encounter), C.c_uint64]),
    "rp_mccfr_import": (C.c_int, [C.c_void_p, C.POINTER(Encounter), C.c_uint64, C.c_uint64]),
    "rp_mccfr_policy": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "rp_mccfr_exploitability": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "rp_mccfr_sum_regret": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "rp_mccfr_set_batch": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rp_mccfr_set_update_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "rp_mccfr_set_rng": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(HashStreams)]),
    "rp_game_hash_streams": (C.c_int, [C.c_void_p, C.POINTER(HashStreams)]),
    "rp_mccfr_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rp_mccfr_traversal_variant": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "rp_game_skeleton": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "rp_mccfr_set_shard": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "rp_mccfr_summary_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "rp_mccfr_step_local": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rp_mccfr_step_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "rp_mccfr_window_local": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "rp_mccfr_window_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "rp_comm_unique_id": (C.c_int, [C.c_void_p]),
    "rp_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rp_comm_adopt": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rp_comm_destroy": (C.c_int, [C.c_void_p]),
    "rp_mccfr_step_comm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "rp_kmeans_step_comm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "rp_mccfr_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "rp_mccfr_kernel_time": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "rp_profile_create": (C.c_int, [C.c_int, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.POINTER(Hyper), C.c_void_p, C.c_uint32,
                                    C.POINTER(C.c_void_p)]),
    "rp_profile_destroy": (C.c_int, [C.c_void_p]),
    "rp_profile_apply": (C.c_int, [C.c_void_p, C.POINTER(Decisions), C.c_int]),
    "rp_profile_sync": (C.c_int, [C.c_void_p]),
    "rp_profile_epoch": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rp_profile_set_epoch": (C.c_int, [C.c_void_p, C.c_uint64]),
    "rp_profile_get_rows": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "rp_profile_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rp_profile_entry_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "rp_profile_summarize": (C.c_int, [C.c_void_p, C.POINTER(Decisions), C.c_void_p, C.POINTER(C.c_uint32)]),
    "rp_profile_fold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "rp_profile_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "rp_profile_kernel_time": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "rp_sinkhorn_hp_default": (None, [C.POINTER(SinkhornHP)]),
    "rp_kmeans_create": (
        C.c_int,
        [C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(SinkhornHP), C.c_uint64,
         C.c_int, C.POINTER(C.c_void_p)],
    ),
    "rp_kmeans_create_device": (
        C.c_int,
        [C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(SinkhornHP), C.c_uint64,
         C.c_int, C.POINTER(C.c_void_p)],
    ),
    "rp_kmeans_destroy": (C.c_int, [C.c_void_p]),
    "rp_kmeans_init_centroids": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rp_kmeans_set_centroids": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rp_kmeans_set_centroid": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "rp_kmeans_get_point": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "rp_kmeans_kpp_begin": (C.c_int, [C.c_void_p]),
    "rp_kmeans_kpp_total": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rp_kmeans_kpp_pick": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "rp_kmeans_kpp_update": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rp_kmeans_init_bounds": (C.c_int, [C.c_void_p]),
    "rp_kmeans_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "rp_kmeans_step_naive": (C.c_int, [C.c_void_p]),
    "rp_kmeans_assign": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rp_kmeans_bounds": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rp_kmeans_centroids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rp_kmeans_metric": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rp_kmeans_rms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "rp_kmeans_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rp_kmeans_exp_evals": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rp_kmeans_prune_stats": (C.c_int, [C.c_void_p, C.POINTER(PruneStats)]),
    "rp_kmeans_bound_intervals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rp_kmeans_kpp_bound_probe": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "rp_kmeans_kpp_bound_probe_at": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_void_p]),
    "rp_kmeans_refresh_stats": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rp_kmeans_upper_interval": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rp_kmeans_pairwise_last": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rp_weighted_index_probe": (C.c_int, [C.c_int, C.c_uint64, C.c_void_p, C.c_float, C.c_int, C.c_void_p]),
    "rp_kmeans_stats_ex": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rp_kmeans_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rp_kmeans_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "rp_kmeans_kernel_time": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "rp_kmeans_partial_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "rp_kmeans_step_local": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rp_kmeans_step_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "rp_sinkhorn_set_libm": (C.c_int, [C.c_int]),
    "rp_sinkhorn_divergence": (
        C.c_int,
        [C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SinkhornHP), C.c_int, C.c_void_p],
    ),
    "rp_sinkhorn_cost": (
        C.c_int,
        [C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SinkhornHP), C.c_int, C.c_void_p,
         C.c_void_p],
    ),
    "rp_sinkhorn_flow": (C.c_int, [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SinkhornHP), C.c_int, C.c_void_p, C.c_void_p]),
    "rp_equity_variation": (C.c_int, [C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "rp_mccfr_train": (C.c_int, [C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "rp_nlhe_create": (C.c_int, [C.c_int, C.c_uint32, C.c_int, C.c_int, C.POINTER(Hyper), C.c_uint64, C.c_uint32, C.c_void_p,
                                 C.POINTER(C.c_void_p)]),
    "rp_nlhe_destroy": (C.c_int, [C.c_void_p]),
    "rp_nlhe_step": (C.c_int, [C.c_void_p, C.c_int]),
    "rp_nlhe_set_sampling": (C.c_int, [C.c_void_p, C.c_int]),
    "rp_nlhe_set_rng": (C.c_int, [C.c_void_p, C.c_int]),
    "rp_nlhe_set_exact": (C.c_int, [C.c_void_p, C.c_int]),
    "rp_kmeans_set_rng": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rp_kmeans_set_libm": (C.c_int, [C.c_void_p, C.c_int]),
    "rp_kmeans_set_prune": (C.c_int, [C.c_void_p, C.c_int]),
    "rp_kmeans_prune_stats_sized": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rp_nlhe_train": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_char_p, C.c_size_t]),
    "rp_nlhe_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "rp_nlhe_kernel_time": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "rp_nlhe_census": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rp_nlhe_last_shape": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rp_nlhe_batch": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)] + [C.c_void_p] * 9),
    "rp_nlhe_epoch": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rp_nlhe_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rp_nlhe_export": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rp_nlhe_import": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "rp_nlhe_set_shard": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "rp_nlhe_entry_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]),
    "rp_nlhe_step_local": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "rp_nlhe_step_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "rp_nlhe_step_comm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "rp_nlhe_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rp_nlhe_sync": (C.c_int, [C.c_void_p]),
    "rp_profile_set_rows": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "rp_nlhe_playouts": (C.c_int, [C.c_int, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rp_hand_strength": (C.c_int, [C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]),
    "rp_obs_canonical": (C.c_int, [C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]),
    "rp_isomorphisms": (C.c_int, [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "rp_river_equity": (C.c_int, [C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rp_lookup_create": (C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "rp_lookup_destroy": (C.c_int, [C.c_void_p]),
    "rp_lookup_get": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "rp_lookup_project": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "rp_deuce_kernel_ms": (C.c_int, [C.POINTER(C.c_double)]),
    "rp_pgcopy_write": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "rp_pgcopy_read": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "rp_artifact_write_lookup": (C.c_int, [C.c_char_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]),
    "rp_artifact_write_metric": (C.c_int, [C.c_char_p, C.c_int, C.c_uint32, C.c_void_p]),
    "rp_artifact_write_blueprint": (C.c_int, [C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.POINTER(C.c_uint64)]),
    "rp_artifact_write_transitions": (C.c_int, [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
}

_lib = None


def load() -> C.CDLL:
    """Load librp_mi355x.so (once).  Raises if it has not been built: no silent fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C robopoker_amd/csrc).  The MI355X path has no CPU fallback."
        )
    # One HIP runtime per process: PyTorch ships its own libamdhip64, and whichever copy initialises the device first owns it —
    # a torch imported AFTER the library's first HIP call finds "No HIP GPUs".  Tests, bench.py and the multi-GPU harness all use
    # torch tensors for device buffers, so when torch is installed it is imported first, whatever order the callers import in.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != RP_OK:
        raise RpError(code, load().rp_last_error().decode("utf-8", "replace"))


def declared_symbols() -> list[str]:
    return sorted(_SIGNATURES)
