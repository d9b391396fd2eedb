# Round evidence for profiles/rNN: bench line, rocprofv3 kernel stats and HBM-traffic counters for the three storages
# of BASELINE config 3, plus the Ns=18 product, config 5 and an ed_total_ud=F sector.
# Usage (on the GPU box): bash scripts/profile_round.sh ; outputs under gpurun_out/prof_round/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
python3 bench.py --steps 20 --warmup 3 > $OUT/bench_c3_csr_plain.json 2> $OUT/bench_plain.err
echo "done plain bench"
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
# three-pass iteration for comparison (EDGPU_LZ_FUSED=0)
EDGPU_LZ_FUSED=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_kron_unfused -- python3 bench.py --steps 20 --warmup 3 --no-extra --no-cpu --storage kron > $OUT/bench_c3_kron_unfused_traced.json 2> $OUT/trace_kron_unfused.err
cp $(find $OUT/trace_kron_unfused -name "*kernel_stats.csv" | head -1) $OUT/bench_c3_kron_unfused_kernel_stats.csv
# Ns=18 (BASELINE config 4) on one GPU: product kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ns18 -- python3 scripts/microbench.py 17 direct 3 > $OUT/ns18_direct_product.txt 2>&1
cp $(find $OUT/trace_ns18 -name "*kernel_stats.csv" | head -1) $OUT/ns18_direct_kernel_stats.csv
for pmc in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_ns18_$pmc -- python3 scripts/microbench.py 17 direct 1 > $OUT/pmc_ns18_$pmc.log 2>&1
done
python3 scripts/pmc_summary.py $OUT 2>/dev/null | grep -A3 "pmc_ns18" > $OUT/ns18_pmc_summary.txt
python3 scripts/microbench.py 17 kron 3 lanczos > $OUT/ns18_kron_iteration.txt 2>&1
echo "done ns18"
# config 5 and ed_total_ud=F
for st in kron direct csr; do python3 scripts/bench_c5.py 7 $st; done > $OUT/c5_times.txt 2>&1
python3 scripts/microbench.py 15 csr 5 > $OUT/c3_isolated_products.txt 2>&1
python3 scripts/microbench.py 15 kron 10 >> $OUT/c3_isolated_products.txt 2>&1
python3 scripts/microbench.py 15 direct 10 >> $OUT/c3_isolated_products.txt 2>&1
for st in kron direct csr; do python3 scripts/microbench_orbs.py 7 $st 10; done > $OUT/orbs_times.txt 2>&1
EDGPU_COMPOSITE=0 python3 scripts/microbench_orbs.py 7 kron 10 >> $OUT/orbs_times.txt 2>&1
python3 scripts/gs_time.py 15 kron > $OUT/gs_time_kron.txt 2>&1
python3 scripts/bw_probe.py > $OUT/bw_probe.txt 2>&1
python3 scripts/small_sector_bench.py > $OUT/small_sectors.txt 2>&1
python3 scripts/eig_time.py 15 kron 2 > $OUT/eig_time_c3.txt 2>&1
python3 scripts/create_time.py > $OUT/create_time.txt 2>&1
cat $OUT/small_sectors.txt $OUT/eig_time_c3.txt $OUT/create_time.txt $OUT/c5_times.txt $OUT/orbs_times.txt $OUT/ns18_direct_product.txt $OUT/ns18_kron_iteration.txt $OUT/c3_isolated_products.txt $OUT/gs_time_kron.txt $OUT/bw_probe.txt | grep -v "^W2026\|^E2026"
