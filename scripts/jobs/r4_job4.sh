#!/bin/bash
# round 4: kernel traces -- fused CSR iteration at c3, emulated rank 3 of 8 (Ns=18 direct, Ns=16 csr), Ns=18 one-GPU product; sector sweep
mkdir -p gpurun_out/r4p
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for f in 1 0; do
  EDGPU_CSR_FUSED=$f rocprofv3 --kernel-trace --stats -d gpurun_out/r4p/csr_fused$f -o t -- python3 bench.py --no-cpu --no-extra --steps 20 --warmup 3 > gpurun_out/r4p/csr_fused$f.json 2>gpurun_out/r4p/csr_fused$f.err
done
rocprofv3 --kernel-trace --stats -d gpurun_out/r4p/emu_ns18_r3 -o t -- python3 bench.py --emulate-rank 3 --of 8 --nbath 17 --storage direct --steps 10 --warmup 2 > gpurun_out/r4p/emu_ns18_r3.json 2>gpurun_out/r4p/emu.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r4p/emu_ns16csr_r3 -o t -- python3 bench.py --emulate-rank 3 --of 8 --nbath 15 --storage csr --steps 20 --warmup 3 > gpurun_out/r4p/emu_ns16csr_r3.json 2>>gpurun_out/r4p/emu.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r4p/ns18_direct -o t -- python3 scripts/microbench.py 17 direct 10 > gpurun_out/r4p/ns18_direct.txt 2>&1
find gpurun_out/r4p -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -12 "$f" | cut -c1-200; done
python scripts/sector_sweep.py 11 kron 1 > gpurun_out/r4p/sweep_ns12_kron.txt 2>&1; head -3 gpurun_out/r4p/sweep_ns12_kron.txt
python scripts/sector_sweep.py 7 kron 1 > gpurun_out/r4p/sweep_ns8_kron.txt 2>&1; head -3 gpurun_out/r4p/sweep_ns8_kron.txt
