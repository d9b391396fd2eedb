#!/bin/bash
# Copy the outputs of scripts/profile_round.sh (gpurun_out/prof_round, merged back by gpurun) into the tracked evidence of the
# round: profiles/rNN/ and profiles/pmc_traffic.json.  eig_time_c3.txt keeps its history: the new line is appended with a note.
# usage: scripts/collect_profile_round.sh r03 "note for the eigensolver line"
set -e
cd "$(dirname "$0")/.."
R=${1:-r03}; NOTE=${2:-profile round}
O=gpurun_out/prof_round
for f in bench_c3_csr_plain.json bench_c3_csr_traced.json bench_c3_kron_traced.json bench_c3_direct_traced.json bench_c3_kron_unfused_traced.json \
         bench_c3_csr_kernel_stats.csv bench_c3_kron_kernel_stats.csv bench_c3_direct_kernel_stats.csv bench_c3_kron_unfused_kernel_stats.csv \
         pmc_traffic_summary.txt c3_isolated_products.txt gs_time_kron.txt create_time.txt small_sectors.txt bw_probe.txt \
         ns18_direct_product.txt ns18_direct_kernel_stats.csv ns18_pmc_summary.txt ns18_kron_iteration.txt c5_times.txt orbs_times.txt; do
  cp $O/$f profiles/$R/$f
done
cp $O/pmc_traffic.json profiles/pmc_traffic.json
printf '## %s\n%s\n' "$NOTE" "$(grep thick-restart $O/eig_time_c3.txt)" >> profiles/$R/eig_time_c3.txt
python3 - <<PY
import json, sys
sys.path.insert(0, '.')
import bench
d = json.loads(open('profiles/$R/bench_c3_csr_plain.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'eigsh_c3_s')}, 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'], 'cpu', d['cpu_baseline']['value'])
print('stamp', bench.source_stamp(), [r.get('source_stamp') for r in json.load(open('profiles/pmc_traffic.json'))][:1])
PY
grep "kernel time\|total" profiles/$R/pmc_traffic_summary.txt
