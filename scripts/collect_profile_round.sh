#!/bin/bash
# Copy the outputs of scripts/profile_round.sh (gpurun_out/prof_round, merged back by gpurun) into the tracked evidence of the
# round: profiles/rNN/ and profiles/pmc_traffic.json.
# usage: scripts/collect_profile_round.sh r04
set -e
cd "$(dirname "$0")/.."
R=${1:-r04}
O=gpurun_out/prof_round
mkdir -p profiles/$R
for f in bench_c3_csr_plain.json bench_c3_csr_traced.json bench_c3_kron_traced.json bench_c3_direct_traced.json bench_c3_kron_unfused_traced.json \
         bench_c3_csr_kernel_stats.csv bench_c3_kron_kernel_stats.csv bench_c3_direct_kernel_stats.csv bench_c3_kron_unfused_kernel_stats.csv \
         pmc_traffic_summary.txt gs_time_kron.txt create_time.txt small_sectors.txt eig_time_c3.txt gf_leg_c3.txt \
         ns18_direct_product.txt ns18_direct_kernel_stats.csv ns18_pmc_summary.txt ns18_direct_iteration.txt ns18_kron_iteration.txt c5_times.txt orbs_times.txt \
         sector_sweep_ns12.txt sector_sweep_ns8.txt emu_pmc_summary.txt emu_ns18_r3_kernel_stats.csv emu_ns16csr_r3_kernel_stats.csv \
         emu_ns18_r0.json emu_ns18_r3.json emu_ns18_r7.json emu_ns16csr_r0.json emu_ns16csr_r3.json emu_ns16csr_r7.json; do
  if [ -f $O/$f ]; then cp $O/$f profiles/$R/$f; else echo "missing $f"; fi
done
cp $O/pmc_traffic.json profiles/pmc_traffic.json
python3 - <<PY
import json, sys
sys.path.insert(0, '.')
import bench
d = json.loads(open('profiles/$R/bench_c3_csr_plain.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'eigsh_c3_s', 'c4_direct_product_ms', 'c5_kron_iteration_ms')}, 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'], 'cpu', d['cpu_baseline']['value'])
print('stamp', bench.source_stamp(), [r.get('source_stamp') for r in json.load(open('profiles/pmc_traffic.json'))][:1])
PY
grep "kernel time\|total" profiles/$R/pmc_traffic_summary.txt
