# Round evidence for profiles/rNN: bench line, rocprofv3 kernel stats and HBM-traffic counters for the three storages of BASELINE
# config 3, the Ns=18 product, one rank of eight emulated on this GPU
# (Ns=18 on-the-fly, Ns=16 explicit CSR), config 5, ed_total_ud=F, the solvers, the sector sweeps.
# Usage (on the GPU box): bash scripts/profile_round.sh [part]   part: all (default) | bench | ns18 | emu | rest ; outputs under gpurun_out/prof_round/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export PROFILE_ROUND_NAME=r04
PART=${1:-all}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_round
mkdir -p $OUT
trace() { d=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$d -- "$@"; cp $(find $OUT/trace_$d -name "*kernel_stats.csv" | head -1) $OUT/${d}_kernel_stats.csv; }
pmc() { d=$1; shift; for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${d}_$c -- "$@" > $OUT/pmc_${d}_$c.log 2>&1; done; }
if [ $PART = all ] || [ $PART = bench ]; then
  python3 bench.py --steps 20 --warmup 3 > $OUT/bench_c3_csr_plain.json 2> $OUT/bench_plain.err
  echo "done plain bench"
  for st in csr kron direct; do
    trace bench_c3_$st python3 bench.py --steps 20 --warmup 3 --no-extra --no-cpu --storage $st > $OUT/bench_c3_${st}_traced.json 2> $OUT/trace_$st.err
    pmc $st python3 bench.py --steps 5 --warmup 1 --no-extra --no-cpu --storage $st
    echo "done $st"
  done
  python3 scripts/pmc_traffic.py $OUT > $OUT/pmc_traffic_summary.txt
  cat $OUT/pmc_traffic_summary.txt
  # the forms the fused iterations replace
  EDGPU_LZ_FUSED=0 trace bench_c3_kron_unfused python3 bench.py --steps 20 --warmup 3 --no-extra --no-cpu --storage kron > $OUT/bench_c3_kron_unfused_traced.json 2> $OUT/trace_kron_unfused.err
fi
if [ $PART = all ] || [ $PART = ns18 ]; then
  trace ns18_direct python3 scripts/microbench.py 17 direct 3 > $OUT/ns18_direct_product.txt 2>&1
  pmc ns18 python3 scripts/microbench.py 17 direct 1
  python3 scripts/pmc_summary.py $OUT 2>/dev/null | grep -A3 "pmc_ns18" > $OUT/ns18_pmc_summary.txt
  python3 scripts/microbench.py 17 direct 3 lanczos > $OUT/ns18_direct_iteration.txt 2>&1
  python3 scripts/microbench.py 17 kron 3 lanczos > $OUT/ns18_kron_iteration.txt 2>&1
  echo "done ns18"
fi
if [ $PART = all ] || [ $PART = emu ]; then
  for r in 0 3 7; do
    python3 bench.py --emulate-rank $r --of 8 --nbath 17 --storage direct --steps 10 --warmup 2 > $OUT/emu_ns18_r$r.json 2>> $OUT/emu.err
    python3 bench.py --emulate-rank $r --of 8 --nbath 15 --storage csr --steps 20 --warmup 3 > $OUT/emu_ns16csr_r$r.json 2>> $OUT/emu.err
  done
  trace emu_ns18_r3 python3 bench.py --emulate-rank 3 --of 8 --nbath 17 --storage direct --steps 10 --warmup 2 > $OUT/emu_ns18_r3_traced.json 2>> $OUT/emu.err
  trace emu_ns16csr_r3 python3 bench.py --emulate-rank 3 --of 8 --nbath 15 --storage csr --steps 20 --warmup 3 > $OUT/emu_ns16csr_r3_traced.json 2>> $OUT/emu.err
  pmc emu_ns18_r3 python3 bench.py --emulate-rank 3 --of 8 --nbath 17 --storage direct --steps 3 --warmup 1
  pmc emu_ns16csr_r3 python3 bench.py --emulate-rank 3 --of 8 --nbath 15 --storage csr --steps 3 --warmup 1
  python3 scripts/pmc_summary.py $OUT 2>/dev/null | grep -A12 "pmc_emu" > $OUT/emu_pmc_summary.txt
  echo "done emu"
fi
if [ $PART = all ] || [ $PART = rest ]; then
  for st in kron direct csr; do python3 scripts/bench_c5.py 7 $st; done > $OUT/c5_times.txt 2>&1
  for st in kron direct csr; do python3 scripts/microbench_orbs.py 7 $st 10; done > $OUT/orbs_times.txt 2>&1
  python3 scripts/gs_time.py 15 kron > $OUT/gs_time_kron.txt 2>&1
  python3 scripts/small_sector_bench.py > $OUT/small_sectors.txt 2>&1
  python3 scripts/eig_time.py 15 kron 2 > $OUT/eig_time_c3.txt 2>&1
  python3 scripts/create_time.py > $OUT/create_time.txt 2>&1
  python3 scripts/sector_sweep.py 11 kron 1 > $OUT/sector_sweep_ns12.txt 2>&1
  python3 scripts/sector_sweep.py 7 kron 1 > $OUT/sector_sweep_ns8.txt 2>&1
  python3 scripts/gf_leg_time.py kron > $OUT/gf_leg_c3.txt 2>&1
  cat $OUT/small_sectors.txt $OUT/eig_time_c3.txt $OUT/create_time.txt $OUT/c5_times.txt $OUT/orbs_times.txt $OUT/gs_time_kron.txt | grep -v "^W2026\|^E2026\|amdgpu.ids"
  head -3 $OUT/sector_sweep_ns12.txt $OUT/sector_sweep_ns8.txt
fi
