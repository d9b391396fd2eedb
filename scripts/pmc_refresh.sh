# Re-measure the HBM-traffic counters of the three storages on the CURRENT sources (separate --pmc passes, as
# MI355X_MICROARCH.md prescribes) and write gpurun_out/prof_round/pmc_traffic.json + summary.
# bash scripts/pmc_refresh.sh [csr kron direct]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_round
mkdir -p $OUT
for st in ${@:-csr kron direct}; do
  for pmc in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc_${st}_$pmc
    rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_${st}_$pmc -- python3 bench.py --steps 5 --warmup 1 --no-extra --no-cpu --storage $st > $OUT/pmc_${st}_$pmc.json 2> $OUT/pmc_${st}_$pmc.err
  done
  echo "done $st"
done
python3 scripts/pmc_traffic.py $OUT > $OUT/pmc_traffic_summary.txt
cat $OUT/pmc_traffic_summary.txt
