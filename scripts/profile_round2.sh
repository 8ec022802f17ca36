#!/bin/bash
# Round-2 profile artifacts (run through gpurun from the repo root; results in gpurun_out/prof2/, the judged copies are
# committed under profiles/).  Every counter set is its own rocprofv3 run with --kernel-trace only.
#   1 kernel-trace stats of the MCCFR timed loop (bench.py --no-extras)            -> r02_bench_kernel_stats.txt
#   2 FETCH_SIZE / WRITE_SIZE passes of the same command                           -> r02_mccfr_hbm_traffic.json
#   3 kernel-trace stats of the FULL flop k-means (1 286 792 x 32 iterations)      -> r02_lloyd_full_kernel_stats.txt + .json
#   4 MFMA / issue counters of k_sinkhorn_bound on a slice                         -> r02_mfma_bound_counters.txt
#   5 FETCH_SIZE / WRITE_SIZE of the Elkan bound update at full N                  -> r02_lloyd_bounds_hbm_traffic.txt
#   6 kernel-trace stats of the NLHE traversal step                                -> r02_nlhe_kernel_stats.txt
#   7 kernel-trace stats of the sparse profile step (nlhe-synth)                    -> r02_sparse_kernel_stats.txt
#   2b SQ issue / wait / lane counters of the MCCFR kernels                        -> r02_mccfr_sq_counters.txt
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
BENCH="python $REPO/bench.py --no-extras --steps 40 --warmup 5"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
python $REPO/scripts/rocpd_summary.py $(ls $OUT/kt/*.db | head -1) $OUT/${TAG}_bench_kernel_stats.txt "$BENCH" > /dev/null
grep -o '{"metric.*' $OUT/kt.log > $OUT/${TAG}_bench_line_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o pmc -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o pmc -- $BENCH > $OUT/write.log 2>&1
python $REPO/scripts/pmc_traffic.py $OUT/fetch/pmc_counter_collection.csv $OUT/write/pmc_counter_collection.csv \
    $OUT/${TAG}_mccfr_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of: $BENCH; FETCH_SIZE doubled (gfx950), KiB -> bytes" 8388608 composed > /dev/null
# 2b: SQ issue / wait / lane counters of the same command (two more passes)      -> r02_mccfr_sq_counters.txt
( cd $REPO && bash scripts/pmc_sq.sh ${TAG}mccfr $BENCH ) > $OUT/sq.log 2>&1
{ echo "# scripts/pmc_sq.sh: rocprofv3 --pmc <two passes> --kernel-trace of: $BENCH  (sums over all dispatches of each kernel)"; grep -E "^(void )?rp::" $OUT/sq.log; } > $OUT/${TAG}_mccfr_sq_counters.txt
# 3: the full flop configuration
FK="python $REPO/scripts/full_kmeans.py flop 32"
rocprofv3 --kernel-trace --stats -d $OUT/kl -o kl -- $FK > $OUT/${TAG}_full_flop_kmeans.json 2> $OUT/kl.log
python $REPO/scripts/rocpd_summary.py $(ls $OUT/kl/*.db | head -1) $OUT/${TAG}_lloyd_full_kernel_stats.txt "$FK (N=1286792, K=256, bins=256: k-means++, init_bounds, 32 Elkan iterations, lookup)" > /dev/null
# 4: MFMA counters of the bound kernel (names taken from what this rocprofv3 lists)
LIST=$(rocprofv3 -L 2>/dev/null | tr ' ,\t' '\n\n\n' | grep -E '^SQ_' | sort -u)
pick() { for c in "$@"; do echo "$LIST" | grep -qx "$c" && echo -n "$c "; done; }
C1=$(pick SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY)
C2=$(pick SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU)
QB="python $REPO/scripts/quick_bound.py 32768 1"
echo "# counters pass 1: $C1" > $OUT/${TAG}_mfma_bound_counters.txt
echo "# counters pass 2: $C2" >> $OUT/${TAG}_mfma_bound_counters.txt
rocprofv3 --pmc $C1 --kernel-trace --output-format csv -d $OUT/m1 -o pmc -- $QB > $OUT/m1.log 2>&1
rocprofv3 --pmc $C2 --kernel-trace --output-format csv -d $OUT/m2 -o pmc -- $QB > $OUT/m2.log 2>&1
python - <<PY >> $OUT/${TAG}_mfma_bound_counters.txt
import csv, collections
print("# rocprofv3 --pmc <pass> --kernel-trace of: python scripts/quick_bound.py 32768 1  (flop slice N=32768, K=256: init_bounds, 1 Elkan step, lookup); sums over dispatches")
for f in ["$OUT/m1/pmc_counter_collection.csv", "$OUT/m2/pmc_counter_collection.csv"]:
    try:
        rows = list(csv.DictReader(open(f)))
    except Exception as e:
        print(f, e); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if "k_sinkhorn_bound" not in k and "k_neighbor_masked" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        print(k, len(calls[k]), "dispatches", {c: f"{x:.4e}" for c, x in sorted(v.items())})
PY
tail -4 $OUT/m1.log >> $OUT/${TAG}_mfma_bound_counters.txt
# 5: HBM traffic of the Elkan bound update at full N (2 iterations)
BU="python $REPO/scripts/quick_bounds_full.py"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/bf -o pmc -- $BU > $OUT/bf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/bw -o pmc -- $BU > $OUT/bw.log 2>&1
python - <<PY > $OUT/${TAG}_lloyd_bounds_hbm_traffic.txt
import csv, collections
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md) of: python scripts/quick_bounds_full.py")
print("# (N = 1 286 792, K = 256: lower bounds f32[N][K] = 1.318 GB read + 1.318 GB written per k_bounds_update launch = 2.635 GB algorithmic)")
for f, c, mul in (("$OUT/bf/pmc_counter_collection.csv", "FETCH_SIZE", 2.0), ("$OUT/bw/pmc_counter_collection.csv", "WRITE_SIZE", 1.0)):
    tot = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0]
        if "k_bounds_update" not in k: continue
        tot[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in tot: print(c, k, len(n[k]), "launches", f"{tot[k] / len(n[k]) * 1024 * mul / 1e9:.3f} GB per launch")
PY
# 6: the NLHE traversal
NL="python $REPO/bench.py --workload nlhe --steps 3 --warmup 1 --cpu-seconds 0"
rocprofv3 --kernel-trace --stats -d $OUT/kn -o kn -- $NL > $OUT/kn.log 2>&1
python $REPO/scripts/rocpd_summary.py $(ls $OUT/kn/*.db | head -1) $OUT/${TAG}_nlhe_kernel_stats.txt "$NL" > /dev/null
grep -o '{"metric.*' $OUT/kn.log > $OUT/${TAG}_nlhe_bench_line.json
# 7: the row-addressed profile on synthetic NLHE-scale batches (own radix sort / scan / run lengths)
SP="python $REPO/bench.py --workload nlhe-synth --steps 20 --warmup 3 --cpu-seconds 0"
rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks -- $SP > $OUT/ks.log 2>&1
python $REPO/scripts/rocpd_summary.py $(ls $OUT/ks/*.db | head -1) $OUT/${TAG}_sparse_kernel_stats.txt "$SP" > /dev/null
grep -o '{"metric.*' $OUT/ks.log > $OUT/${TAG}_sparse_bench_line.json
ls -la $OUT | head -40
cat $OUT/${TAG}_mfma_bound_counters.txt $OUT/${TAG}_lloyd_bounds_hbm_traffic.txt
head -12 $OUT/${TAG}_lloyd_full_kernel_stats.txt $OUT/${TAG}_bench_kernel_stats.txt $OUT/${TAG}_nlhe_kernel_stats.txt
