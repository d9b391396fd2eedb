#!/bin/bash
# round 4: full GPU suite after the comm-stream priority was taken out again (pack kernel ahead of the local pass instead); small sectors; emulated rank
mkdir -p gpurun_out/r4x
python -m pytest tests -x -q -m gpu > gpurun_out/r4x/t10.log 2>&1; echo "exit $?" >> gpurun_out/r4x/t10.log; tail -4 gpurun_out/r4x/t10.log
python3 scripts/small_sector_bench.py > gpurun_out/r4x/small_sectors.txt 2>&1; grep -v "amdgpu.ids\|warning: environment" gpurun_out/r4x/small_sectors.txt
python3 scripts/sector_sweep.py 11 kron 1 > gpurun_out/r4x/sweep_ns12_kron.txt 2>&1; head -3 gpurun_out/r4x/sweep_ns12_kron.txt
for r in 3; do python3 bench.py --emulate-rank $r --of 8 --nbath 17 --storage direct --steps 10 --warmup 2 > gpurun_out/r4x/emu_ns18_r$r.json 2>gpurun_out/r4x/emu.err; python3 bench.py --emulate-rank $r --of 8 --nbath 15 --storage csr --steps 20 --warmup 3 > gpurun_out/r4x/emu_ns16csr_r$r.json 2>>gpurun_out/r4x/emu.err; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4x/emu_*r?.json")):
    try:
        d=json.load(open(f)); print(f, "ms/step %.3f product %.3f emu-exchange %.3f"%(d["ms_per_step"],d["time_split_ms"]["hxv_product_of_rank"],d["time_split_ms"]["emulated_exchange_on_comm_stream"]))
    except Exception as e: print(f,"failed",e)
PY
