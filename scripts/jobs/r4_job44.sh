#!/bin/bash
# Davidson stores: plain against nontemporal (libedgpu_exp_nt.so = experiments build with -DEDGPU_DVD_NT), same box, alternating
OUT=gpurun_out/r4z; mkdir -p $OUT
L=$GRAFT_REPO_ROOT/dmft-lanc-ed_amd/edgpu
for rep in 1 2; do
  for lib in libedgpu.so libedgpu_exp_nt.so; do
    echo "== $lib"; EDGPU_LIBRARY=$L/$lib timeout -k 10 120 python3 scripts/dvd_time.py 15 kron 2 2>&1 | grep "default.*keep=nev+1 gcol=1 rotfused=1" | head -1
  done
done > $OUT/dvd_nt.txt 2>&1; cat $OUT/dvd_nt.txt
