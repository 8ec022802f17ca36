#!/bin/bash
# SQ issue/wait counters of one command (own run, kernel-trace only).  usage: scripts/pmc_sq.sh <tag> <cmd...>
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d $OUT -o pmc -- "$@" > $OUT/run.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA \
  --kernel-trace --output-format csv -d $OUT/b -o pmc -- "$@" > $OUT/runb.log 2>&1
python - <<PY
import csv,collections
for f in ["$OUT/pmc_counter_collection.csv","$OUT/b/pmc_counter_collection.csv"]:
    try: rows=list(csv.DictReader(open(f)))
    except Exception as e: print(f,e); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); t=collections.defaultdict(float)
    for r in rows:
        k=r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in agg.items():
        if k.startswith('rp::') or k.startswith('void rp::'): print(k,{c:f"{x:.3e}" for c,x in v.items()})
PY
