cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/j8; mkdir -p $O
for r in 1 2; do
BITS=1 bash tools/ab.sh default tools/ab_libs/persist4.so tools/ab_libs/persist3.so tools/ab_libs/stores.so tools/ab_libs/persist4_stores.so 2>&1 | grep -v Warn
done | tee $O/bound.txt
