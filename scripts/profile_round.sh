# Round evidence for profiles/: bench line, rocprofv3 kernel stats and HBM-traffic counters for the three storages
# of BASELINE config 3.  Usage (on the GPU box): bash scripts/profile_round.sh ; outputs under gpurun_out/prof_round/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
python3 bench.py --steps 20 --warmup 3 > $OUT/bench_c3_csr_plain.json 2> $OUT/bench_plain.err
for st in csr kron direct; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$st -- python3 bench.py --steps 20 --warmup 3 --no-extra --no-cpu --storage $st > $OUT/bench_c3_${st}_traced.json 2> $OUT/trace_$st.err
  cp $(find $OUT/trace_$st -name "*kernel_stats.csv" | head -1) $OUT/bench_c3_${st}_kernel_stats.csv
  for pmc in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_${st}_$pmc -- python3 bench.py --steps 5 --warmup 1 --no-extra --no-cpu --storage $st > $OUT/pmc_${st}_$pmc.json 2> $OUT/pmc_${st}_$pmc.err
  done
  echo "done $st"
done
python3 scripts/pmc_traffic.py $OUT > $OUT/pmc_traffic_summary.txt
cat $OUT/pmc_traffic_summary.txt
