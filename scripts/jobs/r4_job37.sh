#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh rest > gpurun_out/prof_round_rest.log 2>&1; tail -40 gpurun_out/prof_round_rest.log
