#!/bin/bash
# Round 6, after the dual exits (MFMA bound, k-means++ filter): scripts/r6_audits.sh (synthetic + real flop layer, both arithmetics, the
# whole abstraction) and the k-means++ filter's full-size audit in both arithmetics.   usage: gpurun --timeout 3000 -- bash scripts/r6_audits2.sh
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6audit
mkdir -p $OUT
cd $REPO
bash scripts/r6_audits.sh
for lm in glibc contract; do
  timeout 600 python scripts/kpp_audit.py 1286792 $lm > $OUT/r06_kpp_audit_$lm.json 2> $OUT/kpp_$lm.err; echo "kpp $lm:"; cut -c1-900 $OUT/r06_kpp_audit_$lm.json; echo
done
