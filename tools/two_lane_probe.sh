#!/bin/bash
# usage (on the GPU box): tools/two_lane_probe.sh
# What a second execution lane per index could buy: the IVF probe (1M x 768, nlist 1024, nprobe 32, B 256) alone, then two processes of it
# at the same time on the one GPU (each with its own index and stream): if the pair's summed queries/s exceeds the single run's, the small
# latency-bound kernels of one search do overlap with the other's.
cd $GRAFT_REPO_ROOT
one() { python tools/ivf_probe.py 1000000 768 cosine mixture 32 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['nprobe32']['auto']['qps'], d['nprobe32']['auto']['ms_per_batch'])"; }
echo "alone: $(one)"
one > /tmp/lane_a.txt & one > /tmp/lane_b.txt & wait
echo "pair:  $(cat /tmp/lane_a.txt) | $(cat /tmp/lane_b.txt)"
