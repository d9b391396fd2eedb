#!/bin/bash
# round 4: after the pack-kernel rewrite, own rows read in place, and the row-crossing pipeline of the fused CSR local block
mkdir -p gpurun_out/r4q
python -m pytest tests/test_gpu_parity.py tests/test_multirank.py -x -q -m gpu -k "fused_iteration or multirank" > gpurun_out/r4q/t5.log 2>&1; echo "exit $?" >> gpurun_out/r4q/t5.log; tail -4 gpurun_out/r4q/t5.log
grep -q "exit 0" gpurun_out/r4q/t5.log || exit 1
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for f in 1 0; do
  EDGPU_CSR_FUSED=$f python3 bench.py --no-cpu --no-extra --steps 20 --warmup 3 > gpurun_out/r4q/csr_fused$f.json 2>gpurun_out/r4q/csr_fused$f.err
  EDGPU_CSR_FUSED=$f python3 bench.py --no-cpu --no-extra --steps 200 --warmup 50 > gpurun_out/r4q/csr_fused${f}_200.json 2>>gpurun_out/r4q/csr_fused$f.err
done
EDGPU_CSR_FUSED=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4q/csr_fused1 -- python3 bench.py --no-cpu --no-extra --steps 20 --warmup 3 > gpurun_out/r4q/csr_fused1_traced.json 2>gpurun_out/r4q/trace.err
for r in 0 3 7; do python3 bench.py --emulate-rank $r --of 8 --nbath 17 --storage direct --steps 10 --warmup 2 > gpurun_out/r4q/emu_ns18_r$r.json 2>gpurun_out/r4q/emu.err; done
for r in 0 3 7; do python3 bench.py --emulate-rank $r --of 8 --nbath 15 --storage csr --steps 20 --warmup 3 > gpurun_out/r4q/emu_ns16csr_r$r.json 2>>gpurun_out/r4q/emu.err; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4q/emu_ns18_r3 -- python3 bench.py --emulate-rank 3 --of 8 --nbath 17 --storage direct --steps 10 --warmup 2 > gpurun_out/r4q/emu_ns18_r3_traced.json 2>>gpurun_out/r4q/trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4q/emu_ns16csr_r3 -- python3 bench.py --emulate-rank 3 --of 8 --nbath 15 --storage csr --steps 20 --warmup 3 > gpurun_out/r4q/emu_ns16csr_r3_traced.json 2>>gpurun_out/r4q/trace.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4q/csr_fused*.json")):
    try:
        d=json.load(open(f)); print(f,"ms_per_step %.3f product_ms %.3f"%(d["ms_per_step"],d["roofline"]["kernel_ms"]))
    except Exception as e: print(f,"failed",e)
for f in sorted(glob.glob("gpurun_out/r4q/emu_*r?.json")):
    try:
        d=json.load(open(f)); print(f, "ms/step %.3f product %.3f emu-exchange %.3f recvMB %.0f busiest %.0f"%(d["ms_per_step"],d["time_split_ms"]["hxv_product_of_rank"],d["time_split_ms"]["emulated_exchange_on_comm_stream"],d["exchange"]["received_MB_per_product"],d["exchange"]["busiest_peer_received_MB"]))
    except Exception as e: print(f,"failed",e)
PY
find gpurun_out/r4q -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -8 "$f" | cut -c1-160; done
