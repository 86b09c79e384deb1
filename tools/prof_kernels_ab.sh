# Per-kernel times of one script under several builds of libadx (rocprofv3 kernel stats, one GPU, two alternating rounds):
#   PROG=tools/train_time.py PATTERN='hs3x3q_kernel<true, 2>|hs3x3_kernel<0, 2' bash tools/prof_kernels_ab.sh lib1.so lib2.so ...
# prints calls, total ns, average ns of every kernel whose name matches PATTERN (default: the fused stem + pool of faithful_only.py)
export HIP_FORCE_DEV_KERNARG=1
PROG=${PROG:-tools/faithful_only.py}
PATTERN=${PATTERN:-stem_kernel}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 2; do
for lib in "$@"; do
  export ADX_LIB=$GRAFT_REPO_ROOT/$lib
  v=$(basename $lib .so)_$r
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab_$v -- python3 $PROG > gpurun_out/prof_ab_$v.log 2>&1
  f=$(find gpurun_out/prof_ab_$v -name "*kernel_stats.csv" | head -1)
  echo "== $lib"
  grep -E "$PATTERN" $f | cut -d, -f1-4 | sed 's/(adx::[A-Za-z0-9]*)//' | cut -c1-150
done
done
