#!/bin/bash
# Round 5: the tail policy (evaluation ahead of the search once few slots play) on cfg5's share and on the bench's complete run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05o
run() { tag=$1; shift; env "$@" timeout 400 python tools/arena_share.py > gpurun_out/r05o/$tag.json 2>> gpurun_out/r05o/err.txt; python -c "
import json; d=json.loads(open('gpurun_out/r05o/$tag.json').read().strip().splitlines()[-1]); w=d['whole_share']; print('arena $tag', round(w['seconds'],1), 's', w['steps'], 'steps', round(w['sims_per_s']/1e6,2), 'M sims/s', 'ahead', w['rows_evaluated_ahead'])"; }
whole() { tag=$1; shift; env "$@" timeout 400 python tools/tail_knobs.py --extra-steps 0 --cpu-seconds 0 --profile-steps 0 > gpurun_out/r05o/whole_$tag.json 2>> gpurun_out/r05o/err.txt; python -c "
import json; d=json.loads(open('gpurun_out/r05o/whole_$tag.json').read().strip().splitlines()[-1]); w=d['whole_run']; print('selfplay $tag', round(d['value']/1e6,3), 'M', round(w['seconds'],2), 's eff', round(w['efficiency_vs_steady_state'],4), w['steps'], 'steps')"; }
whole default A=1
whole rows1024 TAIL_PREFETCH_ROWS=1024
whole rows1024_share2 TAIL_PREFETCH_ROWS=1024 TAIL_PREFETCH_SHARE=2
whole rows2048 TAIL_PREFETCH_ROWS=2048
whole default_b A=1
run rows1024_share2 TAIL_PREFETCH_ROWS=1024 TAIL_PREFETCH_SHARE=2
run rows2048 TAIL_PREFETCH_ROWS=2048
run rows2048_share2 TAIL_PREFETCH_ROWS=2048 TAIL_PREFETCH_SHARE=2
