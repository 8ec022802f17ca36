#!/bin/bash
# Round 6, the shipped tree: the whole -m gpu suite + the default bench line (r6_final.sh), then the k-means++ filter's full-size audit in both
# arithmetics (the rounds' tripwire on: its sampled claims are checked by the solves) and the synthetic MFMA audit in the reference's arithmetic.
TAG=${1:-r06u}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r6audit
mkdir -p $OUT
cd $REPO
bash scripts/r6_final.sh $TAG
for lm in glibc contract; do
  timeout 600 python scripts/kpp_audit.py 1286792 $lm > $OUT/${TAG}_kpp_audit_$lm.json 2> $OUT/kpp_$lm.err; echo "kpp $lm:"; cut -c1-900 $OUT/${TAG}_kpp_audit_$lm.json; echo
done
RP_AUDIT_LIBM=glibc timeout 900 python scripts/mfma_audit.py synthetic 256 > $OUT/${TAG}_glibc_audit.json 2> $OUT/audit_glibc.err; cut -c1-700 $OUT/${TAG}_glibc_audit.json; echo
