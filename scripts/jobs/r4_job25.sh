#!/bin/bash
OUT=gpurun_out/r4z; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --hip-trace --stats --output-format csv -d $OUT/hiptrace_sweep -- python3 scripts/sector_sweep.py 11 kron 1 > $OUT/sweep_hiptrace.log 2>&1
f=$(ls $OUT/hiptrace_sweep/*/*hip_api_stats.csv | head -1); head -15 $f | cut -c1-200
t=$(ls $OUT/hiptrace_sweep/*/*hip_api_trace.csv | head -1)
python3 - $t <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
big=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]),r["Function"],int(r["Start_Timestamp"])) for r in rows]
big.sort(reverse=True)
t0=min(b[2] for b in big)
for d,f,s in big[:40]: print("%9.2f ms  %-34s at %.1f ms"%(d/1e6,f,(s-t0)/1e6))
PY
rm -f $t   # large
grep "^#" $OUT/sweep_hiptrace.log
