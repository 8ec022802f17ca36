#!/bin/bash
# Round 6, after the last change to the MFMA bound: the full-size audits in both arithmetics (synthetic flop layer: every point of
# init_bounds and lookup against the unpruned search; the REAL flop layer with RP_LLOYD_AUDIT=1 and with / without the refresh bound),
# then the whole abstraction pipeline on the real point sets.   usage: gpurun --timeout 2700 -- bash scripts/r6_audits.sh
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6audit
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
for lm in glibc contract; do
  RP_AUDIT_LIBM=$lm timeout 900 python scripts/mfma_audit.py synthetic 256 > $OUT/r06_${lm}_audit.json 2> $OUT/audit_$lm.err; echo "synthetic $lm:"; cut -c1-700 $OUT/r06_${lm}_audit.json; echo
done
RP_FULL_LIBM=glibc RP_FULL_RNG=reference timeout 900 python scripts/r6_real_audit.py > $OUT/r06_real_flop_audit_glibc.json 2> $OUT/real_glibc.err; echo "real glibc:"; cut -c1-900 $OUT/r06_real_flop_audit_glibc.json; echo
timeout 900 python scripts/r6_real_audit.py > $OUT/r06_real_flop_audit_contract.json 2> $OUT/real_contract.err; echo "real contract:"; cut -c1-900 $OUT/r06_real_flop_audit_contract.json; echo
timeout 600 python scripts/full_abstraction.py > $OUT/r06_full_abstraction.json 2> $OUT/abs.err; echo "abstraction:"; tail -c 900 $OUT/r06_full_abstraction.json; echo
