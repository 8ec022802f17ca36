#!/usr/bin/env python3
"""What binds the bit-faithful Sinkhorn kernels: the rocprofv3 SQ passes of a flop-layer slice (scripts/r5_lloyd_prof.sh) reduced to VALU
issue rates per kernel.   usage: valu_ceiling.py <sq_counters.json> <slice_kernel_us.json> <out.json>

SQ_INSTS_VALU counts a packed-f32 instruction once, the pipe spends two issue slots on it (profiles/r01_valu_issue_rates.txt); the softmin
loop is 106 instructions per 8 terms of which 44 are packed, i.e. 155 plain-equivalent: weight 1.46.  Rates are against the chip's sustained
plain wave64 VALU rate measured by the same micro-benchmark (8.2e11 /s)."""
import json
import sys

SUSTAINED = 8.2e11
WEIGHT = 155.0 / 106.0
cnt = json.load(open(sys.argv[1]))["kernels"]
us = json.load(open(sys.argv[2]))
rows = {}
for name, v in cnt.items():
    c = v["counters"]
    t = us.get(name, {}).get("total_us")
    if not t or c.get("SQ_INSTS_VALU", 0) < 1e9 or "sinkhorn_bound" in name:
        continue
    wc = c["SQ_WAVE_CYCLES"]
    rate = c["SQ_INSTS_VALU"] / (t * 1e-6)
    rows[name.split("::")[-1]] = {
        "dispatches": v["dispatches"], "kernel_ms": round(t / 1e3, 2), "SQ_INSTS_VALU": c["SQ_INSTS_VALU"],
        "valu_instructions_per_s": rate, "plain_equivalent_per_s": rate * WEIGHT, "frac_of_sustained_issue": rate * WEIGHT / SUSTAINED,
        "waves_in_flight": wc * 4 / (t * 1e-6 * 2.4e9),
        # the fraction of SIMD-cycles in which a VALU instruction is issuing, whatever the instruction mix (f64 / packed / plain):
        # SQ_ACTIVE_INST_VALU is per wavefront-cycle in 4-cycle quanta like SQ_WAVE_CYCLES, so busy x waves in flight / 1024 SIMDs
        "simd_valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] / wc * (wc * 4 / (t * 1e-6 * 2.4e9)) / 1024.0,
        "per_wave_cycle": {"valu_busy": c["SQ_ACTIVE_INST_VALU"] / wc, "scalar_busy": c["SQ_ACTIVE_INST_SCA"] / wc,
                           "lds_busy": c["SQ_ACTIVE_INST_LDS"] / wc, "vmem_busy": c.get("SQ_ACTIVE_INST_VMEM", 0) / wc,
                           "wait_any": c["SQ_WAIT_ANY"] / wc, "wait_inst_any": c["SQ_WAIT_INST_ANY"] / wc},
        "lds_bank_conflict_per_lds_cycle": c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_ACTIVE_INST_LDS"], 1)}
doc = {"source": [sys.argv[1], sys.argv[2]],
       "reading": "valu_busy x waves_in_flight / 1024 SIMDs is the fraction of SIMD-cycles a VALU instruction is issuing: the kernels with "
                  "thousands of wavefronts in flight (k_pairwise, k_refresh_pairs, k_neighbor_masked, k_point_dist) keep the VALU pipes "
                  "80 - 97 % busy; wait_inst_any is wavefronts queueing for the busy pipe, not a stall to remove.  LDS, scalar and memory "
                  "pipes are 1 - 7 % busy, bank conflicts are absent.  The kernels with few wavefronts in flight (k_neighbor on a sample "
                  "list, k_kpp_update on the few points with > 32 bins) are launch-shape bound, not instruction bound.",
       "sustained_plain_wave64_valu_per_s": SUSTAINED, "packed_weight": WEIGHT, "kernels": rows}
json.dump(doc, open(sys.argv[3], "w"), indent=1)
for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["kernel_ms"]):
    print(f"{k:28s} {r['kernel_ms']:9.1f} ms  issue {r['frac_of_sustained_issue']:.2f}  waves {r['waves_in_flight']:7.0f}  valu_busy/wave {r['per_wave_cycle']['valu_busy']:.3f}  SIMD busy {r['simd_valu_busy_frac']:.2f}")
