#!/bin/bash
# experiments build, timing-only ablations of k_dwpass2 (WRONG results by construction): what bounds the dw pass?
export EDGPU_LIBRARY=$GRAFT_REPO_ROOT/dmft-lanc-ed_amd/edgpu/libedgpu_exp.so
OUT=gpurun_out/r4z; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for nb in 15 17; do
  st=direct; n=5; [ $nb = 15 ] && n=10
  for abl in 0 1 4 2 8 3 6 10 11 14; do
    export EDGPU_DW_ABL=$abl
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/abl_${nb}_$abl -- python3 scripts/microbench.py $nb $st $n > $OUT/abl_${nb}_$abl.log 2>&1
    f=$(ls $OUT/abl_${nb}_$abl/*/*kernel_stats.csv | head -1)
    echo "nbath=$nb abl=$abl  $(grep k_dwpass2 $f | cut -d, -f2-4)"
    rm -rf $OUT/abl_${nb}_$abl
  done
done
