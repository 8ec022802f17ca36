#!/usr/bin/env python3
"""Per-kernel instruction census of a gfx950 assembly file (hipcc --cuda-device-only -S): which kernels hold f64 exponentials, LDS
table reads / the table initialisation, vector-memory gathers, branches, packed-f32 instructions, registers.
usage: isa_summary.py <file.s> [name filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
names = re.findall(r"^(_Z\w+):\s*; @", txt, re.M)
dem = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()))
print(f"{'kernel':58s} {'instr':>6s} {'f64':>5s} {'rndne':>5s} {'ds_r64':>6s} {'ds_w64':>6s} {'gl_x2':>5s} {'branch':>6s} {'pk':>4s} {'vgpr':>4s} {'lds':>6s}")
for n in names:
    i = txt.index("\n" + n + ":")
    j = re.compile(r"^\.Lfunc_end\d+:", re.M).search(txt, i).start()
    body = txt[i:j]
    d = dem[n].split("(")[0].replace("void ", "")
    if flt not in d or "\t.amdhsa_kernel " + n + "\n" not in txt:
        continue
    ins = [l.split()[0] for l in body.splitlines() if l.strip() and l.startswith("\t") and not l.startswith("\t;") and not l.startswith("\t.")]
    c = lambda p: sum(1 for x in ins if re.match(p, x))
    vg = re.search(r"\.set " + re.escape(n) + r"\.num_vgpr, (\d+)", txt)
    k = txt.index("\t.amdhsa_kernel " + n + "\n")
    ld = re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", txt[k:k + 3000])
    f64 = c(r"v_\w+_f64")
    print(f"{d[:58]:58s} {len(ins):6d} {f64:5d} {c('v_rndne_f64'):5d} {c('ds_read_b64'):6d} {c('ds_write_b64'):6d} {c('global_load_dwordx2'):5d} "
          f"{c('s_cbranch'):6d} {c('v_pk_'):4d} {vg.group(1) if vg else '?':>4s} {ld.group(1) if ld else '?':>6s}")
