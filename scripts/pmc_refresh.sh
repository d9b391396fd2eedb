# Refresh the counter summary bench.py quotes (profiles/pmc_traffic.json) after a change under csrc/: the bench part of the profile
# round only.  On the GPU box: bash scripts/pmc_refresh.sh ; then locally: bash scripts/collect_profile_round.sh r04
bash scripts/profile_round.sh bench
