#!/bin/bash
# Round 6: the tail policy (evaluation ahead of the search once few slots play) on cfg5's share with concurrent arena games (4 096 slots, 3 parts)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_tail; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 400 python tools/arena_share.py > $O/$tag.json 2>> $O/err.txt; python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); w=d['whole_share']; print('arena $tag', round(w['seconds'],1), 's', w['steps'], 'steps', round(w['sims_per_s']/1e6,2), 'M sims/s', 'ahead', w['rows_evaluated_ahead'])"; }
run default A=1
run share2 TAIL_PREFETCH_SHARE=2
run share3 TAIL_PREFETCH_SHARE=3
run rows2048 TAIL_PREFETCH_ROWS=2048
run rows2048_share2 TAIL_PREFETCH_ROWS=2048 TAIL_PREFETCH_SHARE=2
run sims24 TAIL_PREFETCH_SIMS=24
run default_b A=1
