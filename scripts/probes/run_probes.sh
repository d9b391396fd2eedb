# Stream ceilings of a CSR-shaped access pattern (no gathers): bash scripts/probes/run_probes.sh  (binaries built by hipcc in-tree)
cd $GRAFT_REPO_ROOT/scripts/probes
for p in stream_probe stream_probe2; do
  [ -x $p ] || hipcc --offload-arch=gfx950 -O3 -o $p $p.hip
  echo "== $p"; ./$p
done
