#!/bin/bash
# round 5, job 2: graph-cache hygiene tests, tape replay, in-graph RCCL all-reduce at world 1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job2; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_ddp.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
timeout 600 python3 tools/dbg_graph_dist.py nodist > $O/share_nodist.json 2> $O/share_nodist.err
timeout 600 python3 tools/dbg_graph_dist.py dist > $O/share_dist.json 2> $O/share_dist.err
BHNERF_GRAPH_COLLECTIVE=0 timeout 600 python3 tools/dbg_graph_dist.py dist > $O/share_dist_eager_tail.json 2> $O/share_dist_eager_tail.err
tail -15 $O/tests.log; tail -3 $O/share_dist.err; cat $O/share_nodist.json $O/share_dist.json $O/share_dist_eager_tail.json
