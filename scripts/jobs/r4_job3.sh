#!/bin/bash
# round 4: fused explicit-CSR iteration (parity + c3 timing, fused vs three-pass) and the first one-rank-of-eight measurements
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_iteration or c3_full" > gpurun_out/r4_t3.log 2>&1; echo "exit $?" >> gpurun_out/r4_t3.log; tail -4 gpurun_out/r4_t3.log
grep -q "exit 0" gpurun_out/r4_t3.log || exit 1
for f in 1 0; do
  EDGPU_CSR_FUSED=$f python bench.py --no-cpu --no-extra --steps 20 --warmup 3 > gpurun_out/r4_csr_fused$f.json 2>gpurun_out/r4_csr_fused$f.err
  EDGPU_CSR_FUSED=$f python bench.py --no-cpu --no-extra --steps 200 --warmup 50 > gpurun_out/r4_csr_fused${f}_200.json 2>>gpurun_out/r4_csr_fused$f.err
done
python - <<'PY'
import json
for f in ("1","0"):
    for suf in ("","_200"):
        try:
            d=json.load(open("gpurun_out/r4_csr_fused%s%s.json"%(f,suf)))
            print("CSR_FUSED=%s%s ms_per_step %.3f product_ms %.3f forms %s"%(f,suf,d["ms_per_step"],d["roofline"]["kernel_ms"],d["config"]["forms"]))
        except Exception as e: print(f,suf,"failed",e)
PY
for r in 0 3 7; do python bench.py --emulate-rank $r --of 8 --nbath 17 --storage direct --steps 10 --warmup 2 > gpurun_out/r4_emu_ns18_r$r.json 2>gpurun_out/r4_emu.err; done
for r in 0 3 7; do python bench.py --emulate-rank $r --of 8 --nbath 15 --storage csr --steps 20 --warmup 3 > gpurun_out/r4_emu_ns16csr_r$r.json 2>>gpurun_out/r4_emu.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_emu_*.json")):
    try:
        d=json.load(open(f)); print(f, "ms/step %.3f product %.3f emu-exchange %.3f recvMB %.0f busiest %.0f"%(d["ms_per_step"],d["time_split_ms"]["hxv_product_of_rank"],d["time_split_ms"]["emulated_exchange_on_comm_stream"],d["exchange"]["received_MB_per_product"],d["exchange"]["busiest_peer_received_MB"]))
    except Exception as e: print(f,"failed",e)
PY
tail -3 gpurun_out/r4_emu.err
