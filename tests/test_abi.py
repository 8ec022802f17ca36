"""CPU-side checks of the C-ABI: the library loads without a GPU and exports every declared symbol."""
import ctypes as C
import os
import re

import pytest

from robopoker_amd import Game, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_in_header():
    # the C ABI and its diagnostics header (self tests, kernel clocks, traversal census): one library
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("rp_mi355x.h", "rp_mi355x_diag.h"))
    return sorted(set(re.findall(r"RP_API\s+[\w\s\*]+?\b(rp_\w+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_in_header()
    assert len(names) >= 50
    for n in names:
        assert hasattr(lib, n), f"librp_mi355x.so does not export {n}"
    assert sorted(_lib.declared_symbols()) == names, "robopoker_amd/_lib.py and include/rp_mi355x.h disagree"


def test_version_and_device_count_do_not_need_a_gpu():
    lib = _lib.load()
    assert lib.rp_version().decode().startswith("rp_mi355x")
    assert lib.rp_device_count() >= 0


def test_no_cpu_fallback_when_no_device():
    lib = _lib.load()
    if lib.rp_device_count() > 0:
        pytest.skip("a GPU is visible")
    g = Game("kuhn")
    hp = _lib.Hyper()
    lib.rp_hyper_default(C.byref(hp))
    h = C.c_void_p()
    rc = lib.rp_mccfr_create(C.byref(g.table), 3, 1, 0, 8, C.byref(hp), 0, 0, C.byref(h))
    assert rc == _lib.RP_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.rp_last_error()


def test_hyper_defaults_match_reference():
    # mccfr/src/hyperparams/{sampling.rs:39-50, pruning.rs:36-53, training.rs:49-60}
    hp = _lib.Hyper()
    _lib.load().rp_hyper_default(C.byref(hp))
    assert (hp.temperature, hp.smoothing) == (1.0, 2.0)
    assert abs(hp.curiosity - 0.05) < 1e-7 and abs(hp.prune_explore - 0.05) < 1e-7
    assert hp.prune_threshold == -3e5 and hp.prune_warmup == 16384 and hp.regret_min == -4e6
    s = _lib.SinkhornHP()
    _lib.load().rp_sinkhorn_hp_default(C.byref(s))
    assert abs(s.temperature - 0.025) < 1e-9 and s.iterations == 128 and abs(s.tolerance - 5e-4) < 1e-9


def test_game_table_check_rejects_bad_tables():
    lib = _lib.load()
    g = Game("kuhn")
    bad = _lib.GameTable()
    C.memmove(C.byref(bad), C.byref(g.table), C.sizeof(bad))
    bad.train_root = bad.n_states + 5
    assert lib.rp_game_table_check(C.byref(bad)) == _lib.RP_ERR_INVALID
    assert lib.rp_game_info_id(g._h, b"not-an-infoset", C.byref(C.c_uint32())) == _lib.RP_ERR_INVALID


def test_infoset_names_follow_reference_display():
    # Composite Display = "{secret}|{public}" (mccfr/src/state/composite.rs:50-57)
    k = Game("kuhn")
    assert sorted(k.info_name(i) for i in range(12)) == sorted(
        r + "|" + h for r in "JQK" for h in ("", "X", "B", "XB"))
    l = Game("leduc")
    names = {l.info_name(i) for i in range(l.n_infos)}
    assert "J|" in names and "K|XR" in names and "Q|K|XRCXR" in names and "J|J|XX" in names


def _build_c_host(tmp_path):
    import subprocess

    exe = str(tmp_path / "abi_smoke")
    lib_dir = os.path.join(ROOT, "robopoker_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-O1", os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
                           "-o", exe, "-L" + lib_dir, "-l:librp_mi355x.so", "-Wl,-rpath," + lib_dir])
    return exe


def test_header_is_valid_c11_and_a_plain_c_host_links(tmp_path):
    # the drop-in boundary is C, not C++: the header must compile with gcc -std=c11 -Werror and a C program must link
    import subprocess

    exe = _build_c_host(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if _lib.load().rp_device_count() > 0:
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        # no device: the library must refuse (RP_ERR_NO_DEVICE -> exit code 2), never compute on the CPU
        assert r.returncode == 2, r.stdout + r.stderr
        assert "no HIP device" in r.stderr or "NO_DEVICE" in r.stderr or "device" in r.stderr


def test_builtin_games_match_their_compile_time_skeletons():
    # csrc/traverse_static.hpp: the traversal is instantiated over a game's action skeleton when the table matches it node for
    # node for every chance outcome (checked on the host at solver creation).  Kuhn and both Leducs must take that path, a game
    # with three actions (RPS) must not.
    from robopoker_amd import Game

    assert Game("kuhn").skeleton() == "kuhn"
    assert Game("leduc").skeleton() == "leduc"
    assert Game("leduc_wide").skeleton() == "leduc"
    assert Game("rps").skeleton() == ""
