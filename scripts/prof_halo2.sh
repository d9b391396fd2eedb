# Local pass for rows beyond LDS (k_uppass_halo, Ns=18): which stream of the pass it waits for.  EDGPU_ABL bits (results wrong):
# 1 amplitudes constant, 2 gathers read the lane's own element, 4 halo values not loaded, 8 hop words loaded for two rows only, 16 no stores
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_halo2
rm -rf $OUT; mkdir -p $OUT
for a in 0 4 8 12 16 28 31; do
  EDGPU_ABL=$a rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$a -- python3 scripts/microbench.py 17 direct 3 > $OUT/trace_$a.log 2>&1
  echo "EDGPU_ABL=$a" >> $OUT/ablation.txt
  grep "k_uppass_halo\|k_dwpass2" $(find $OUT/trace_$a -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 >> $OUT/ablation.txt
done
cat $OUT/ablation.txt
