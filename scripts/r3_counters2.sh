#!/bin/bash
# which unit is k_nl_expand waiting for?  (separate rocprofv3 --pmc passes, kernel-trace only)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
rocprofv3 -L 2>/dev/null | tr ' ,\t' '\n\n\n' | grep -E '^(TA_|TCP_|TCC_|SQ_)' | sort -u > $OUT/counters.txt
wc -l $OUT/counters.txt
CMD="python $REPO/bench.py --workload nlhe --nlhe-batch 65536 --steps 3 --warmup 2 --cpu-seconds 0"
pass() {
  rm -rf $OUT/run
  rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT/run -o pmc -- $CMD > $OUT/run.log 2>&1
  python - <<PY
import csv,collections
try: rows=list(csv.DictReader(open("$OUT/run/pmc_counter_collection.csv")))
except Exception as e: print("failed: $1", e); rows=[]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k=r['Kernel_Name'].split('(')[0]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if 'k_nl_expand' in k or 'k_nl_children' in k: print(k,{c:f"{x:.3e}" for c,x in v.items()})
PY
}
pass "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
pass "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
pass "SQ_WAIT_INST_LDS SQ_INSTS_GDS SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
pass "TA_BUSY_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum"
pass "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum"
pass "GRBM_GUI_ACTIVE GRBM_COUNT TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum TCP_TCC_CC_READ_REQ_sum TCP_TCC_RW_READ_REQ_sum"
rm -rf $OUT/run
grep -E "^(TA_|TCP_|TCC_).*(BUSY|STALL)" $OUT/counters.txt | head -40
