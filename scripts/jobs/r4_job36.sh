#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh bench > gpurun_out/prof_round_bench.log 2>&1; tail -12 gpurun_out/prof_round_bench.log
