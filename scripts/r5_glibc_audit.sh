#!/bin/bash
# Round 5: the filters in front of the exact solves, audited at FULL size in glibc's arithmetic (VERDICT r4 task 3), the new mode tests,
# and a kernel trace of the contract layer.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5glibc
mkdir -p $OUT
cd $REPO
export RP_FIXTURE_CACHE=/tmp
echo "== tests"; date +%T
timeout 400 python -m pytest tests/test_gpu_z_glibc_mode.py -m gpu -q -x -k "pruned_glibc or set_prune or layer_mode" -p no:cacheprovider 2>&1 | tail -5
echo "== kpp audit, glibc"; date +%T
timeout 300 python scripts/kpp_audit.py 1286792 glibc > $OUT/r05_kpp_audit_glibc.json 2> $OUT/kpp.err; cat $OUT/r05_kpp_audit_glibc.json; tail -2 $OUT/kpp.err
echo "== kernel trace, contract"; date +%T
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
rm -rf $OUT/kt
timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $REPO/scripts/full_kmeans.py flop 32 > $OUT/r05_lloyd_full_line.json 2> $OUT/kt.err
python $REPO/scripts/rocpd_summary.py "$(ls $OUT/kt/*.db 2>/dev/null | head -1)" $OUT/r05_lloyd_full_kernel_stats.txt "python scripts/full_kmeans.py flop 32" | cut -c1-140 | head -24
rm -rf $OUT/kt
cd $REPO
echo "== full-size audit of the pruned passes, glibc"; date +%T
RP_AUDIT_LIBM=glibc timeout 900 python scripts/mfma_audit.py synthetic 256 > $OUT/r05_glibc_audit.json 2> $OUT/audit.err; cat $OUT/r05_glibc_audit.json; tail -3 $OUT/audit.err
date +%T
