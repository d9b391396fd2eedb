# Sustained read rate vs data pattern, with the board power / clocks sampled beside it (read-only rocm-smi queries).
# bash scripts/probes/run_power_probe.sh  -> gpurun_out/power_probe/
cd $GRAFT_REPO_ROOT/scripts/probes
OUT=$GRAFT_REPO_ROOT/gpurun_out/power_probe; mkdir -p $OUT
[ -x stream_probe3 ] || hipcc --offload-arch=gfx950 -O3 -o stream_probe3 stream_probe3.hip
( for i in $(seq 1 60); do echo "t=$(date +%s.%N | cut -c1-12)"; rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk\|socclk" ; sleep 0.4; done ) > $OUT/smi_stream.txt 2>&1 &
SMI=$!
./stream_probe3 > $OUT/stream_probe3.txt 2>&1
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
cat $OUT/stream_probe3.txt
# the same sampler beside the explicit-CSR Lanczos loop (c3, 1500 iterations ~ 15 s)
cd $GRAFT_REPO_ROOT
( for i in $(seq 1 40); do echo "t=$(date +%s.%N | cut -c1-12)"; rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk\|socclk" ; sleep 0.4; done ) > $OUT/smi_csr.txt 2>&1 &
SMI=$!
python3 bench.py --steps 1500 --warmup 3 --no-extra --no-cpu > $OUT/bench_csr_1500.json 2> $OUT/bench_csr_1500.err
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
grep -c "t=" $OUT/smi_stream.txt $OUT/smi_csr.txt
