"""Builds tests/emul/_build/librp_emul.so: the library's kernel SOURCES compiled as plain C++ against the wave64
execution model of tests/emul/ (hip/hip_runtime.h, wavesim.cpp).  TEST INFRASTRUCTURE: it checks kernel logic where
there is no GPU; robopoker_amd/ never loads it.

    python tests/emul/build.py [-j N]

Two spellings of the sources are rewritten on the way in (the product sources are not touched):
  extern __shared__ ... T name[];   ->  T* name = the workgroup's dynamic LDS
  asm volatile("s_waitcnt ...")     ->  a host fence
  comm.cpp's dlopen("librccl.so.1")  ->  tests/emul/fake_rccl.cpp's library
"""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "robopoker_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "librp_emul.so")
CLANG = os.environ.get("RP_EMUL_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++17", os.environ.get("RP_EMUL_OPT", "-O1"), *(["-g"] if os.environ.get("RP_EMUL_DEBUG") else ["-gline-tables-only"]), "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-strict-aliasing",
         "-fvisibility=hidden", "-pthread", "-Wno-unknown-pragmas", "-Wno-unused-function", "-Wno-pass-failed",
         "-Wno-unknown-attributes", "-Wno-unused-value", "-Wno-c++20-extensions"]

EXTERN_SHARED = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][\w:<> ]*?)\s+(\w+)\[\];")
ASM_WAIT = re.compile(r'asm\s+volatile\("s_waitcnt[^"]*"[^;]*;')


def rewrite(text: str, name: str = "") -> str:
    if name == "comm.cpp":  # the library's dlopen of RCCL: the stand-in of tests/emul/fake_rccl.cpp, by absolute path (torch has the
        # real librccl.so.1 mapped already, and dlopen by soname would hand that one back)
        names = '"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"'
        assert names in text
        text = text.replace(names, '"' + os.path.join(OUT, "rccl", "librp_emul_rccl.so") + '"')
    text = EXTERN_SHARED.sub(lambda m: f"{m.group(1)}* {m.group(2)} = reinterpret_cast<{m.group(1)}*>(::emu::dyn_smem());", text)
    text = ASM_WAIT.sub("__atomic_thread_fence(__ATOMIC_SEQ_CST);", text)
    return text


def stage_sources() -> list[str]:
    """copies csrc/ (rewritten) and include/ into _build/src so that relative includes keep working"""
    src_dir = os.path.join(OUT, "src", "robopoker_amd", "csrc")
    inc_dir = os.path.join(OUT, "src", "include")
    os.makedirs(src_dir, exist_ok=True)
    os.makedirs(inc_dir, exist_ok=True)
    units = []
    for name in sorted(os.listdir(CSRC)):
        path = os.path.join(CSRC, name)
        if not os.path.isfile(path) or not name.endswith((".hip", ".hpp", ".cpp", ".h")):
            continue
        new = rewrite(open(path).read(), name)
        dst = os.path.join(src_dir, name)
        if not os.path.exists(dst) or open(dst).read() != new:
            open(dst, "w").write(new)
        if name.endswith((".hip", ".cpp")):
            units.append(dst)
    for name in os.listdir(os.path.join(ROOT, "include")):
        new = open(os.path.join(ROOT, "include", name)).read()
        dst = os.path.join(inc_dir, name)
        if not os.path.exists(dst) or open(dst).read() != new:
            open(dst, "w").write(new)
    return units


def stage_selfcheck() -> str:
    """tests/emul/selfcheck.cpp goes through the same two rewrites as the product sources (it uses dynamic LDS)"""
    dst = os.path.join(OUT, "src", "selfcheck.cpp")
    new = rewrite(open(os.path.join(HERE, "selfcheck.cpp")).read())
    if not os.path.exists(dst) or open(dst).read() != new:
        open(dst, "w").write(new)
    return dst


def digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def compile_one(src: str, obj: str, deps_sig: str) -> None:
    stamp = obj + ".sig"
    sig = deps_sig + digest([src])
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == sig:
        return
    cmd = [CLANG, *FLAGS, "-I", HERE, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-20000:])
        raise SystemExit(f"emul build failed: {os.path.basename(src)}")
    open(stamp, "w").write(sig)


def build(jobs: int = 8, only=None) -> str:
    os.makedirs(OUT, exist_ok=True)
    units = stage_sources()
    if only:
        units = [u for u in units if os.path.basename(u) in only]
    src_dir = os.path.dirname(units[0])
    headers = [os.path.join(src_dir, n) for n in os.listdir(src_dir) if n.endswith((".hpp", ".h"))]
    headers += [os.path.join(OUT, "src", "include", n) for n in os.listdir(os.path.join(OUT, "src", "include"))]
    headers += [os.path.join(HERE, "hip", "hip_runtime.h")]
    deps_sig = digest(headers) + " ".join(FLAGS)
    units.append(os.path.join(HERE, "wavesim.cpp"))
    if not only:
        units.append(stage_selfcheck())
    objs = [os.path.join(OUT, os.path.basename(u) + ".o") for u in units]
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(lambda so: compile_one(so[0], so[1], deps_sig), zip(units, objs)))
    link_sig = digest(objs)
    link_stamp = LIB + ".sig"
    if not (os.path.exists(LIB) and os.path.exists(link_stamp) and open(link_stamp).read() == link_sig) and not only:
        # plain __device__ functions defined in headers: one copy per device image in the product, identical copies here.
        # Linked under another name and renamed: a process that is loading the library never sees a half-written file.
        tmp = LIB + f".{os.getpid()}.tmp"
        cmd = [CLANG, "-shared", "-o", tmp, *objs, "-pthread", "-ldl", "-Wl,--allow-multiple-definition"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-20000:])
            raise SystemExit("emul link failed")
        os.replace(tmp, LIB)
        open(link_stamp, "w").write(link_sig)
    build_fake_rccl()
    return LIB


RCCL_DIR = os.path.join(OUT, "rccl")


def build_fake_rccl() -> str:
    """tests/emul/fake_rccl.cpp -> _build/rccl/librp_emul_rccl.so (ranks = processes of this host); the staged comm.cpp opens it"""
    os.makedirs(RCCL_DIR, exist_ok=True)
    src = os.path.join(HERE, "fake_rccl.cpp")
    lib = os.path.join(RCCL_DIR, "librp_emul_rccl.so")
    stamp = lib + ".sig"
    sig = digest([src])
    if os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read() == sig:
        return lib
    r = subprocess.run([CLANG, "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fvisibility=hidden", "-pthread", src, "-o", lib, "-lrt"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-8000:])
        raise SystemExit("emul build failed: fake_rccl.cpp")
    open(stamp, "w").write(sig)
    return lib


if __name__ == "__main__":
    j = 8
    only = None
    if "-j" in sys.argv:
        j = int(sys.argv[sys.argv.index("-j") + 1])
    if "--only" in sys.argv:
        only = sys.argv[sys.argv.index("--only") + 1].split(",")
    print(build(j, only))
