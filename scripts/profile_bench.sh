# rocprofv3 evidence for bench.py's default run (c3, storage=csr).  Writes under gpurun_out/prof_bench/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
python3 bench.py --steps 20 --warmup 3 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-extra --no-cpu > $OUT/bench_traced.json 2> $OUT/bench_traced.err
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $pmc | tr ' ' '_')
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$tag -- python3 bench.py --steps 5 --warmup 1 --no-extra --no-cpu > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err
done
find $OUT -name "*stats*.csv" | head
