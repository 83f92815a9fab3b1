#!/bin/bash
# Everything profiles/traffic.json and DESIGN.md's tables are made from, on the round's FINAL build (entries are tied to the
# library / source hashes: profiles/make_traffic.py).  usage: bash profiles/collect_round.sh r04 <part>     part = a | b | c | d | e | f (e: the direct / whole sharded legs of a alone; f: all five sharded legs alone)
round=$1; part=$2
case $part in
a) bash profiles/collect.sh $round c2 && bash profiles/collect.sh $round c2-zipf --index-dist zipf && \
   bash profiles/collect.sh $round c4 --workload c4 && bash profiles/collect.sh $round c4-l32 --workload c4 --pooling 32 && \
   bash profiles/collect_dist_pmc.sh $round dist-c4-rows-l1 --workload c4 --rows-scale 0.125 --replicate-mb 64 && \
   bash profiles/collect_dist_pmc.sh $round dist-c4-rows-l32 --workload c4 --rows-scale 0.125 --replicate-mb 64 --pooling 32 && \
   DIST_PMC_DIRECT=1 bash profiles/collect_dist_pmc.sh $round dist-c4-rows-l1-direct --workload c4 --rows-scale 0.125 --replicate-mb 64 && \
   DIST_PMC_DIRECT=1 bash profiles/collect_dist_pmc.sh $round dist-c2-rows-l1-direct --shard-mode rows --replicate-mb 64 && \
   DIST_PMC_DIRECT=1 bash profiles/collect_dist_pmc.sh $round dist-c2-whole-l1 --shard-mode whole --replicate-mb 64 ;;
e) DIST_PMC_DIRECT=1 bash profiles/collect_dist_pmc.sh $round dist-c4-rows-l1-direct --workload c4 --rows-scale 0.125 --replicate-mb 64 && \
   DIST_PMC_DIRECT=1 bash profiles/collect_dist_pmc.sh $round dist-c2-rows-l1-direct --shard-mode rows --replicate-mb 64 && \
   DIST_PMC_DIRECT=1 bash profiles/collect_dist_pmc.sh $round dist-c2-whole-l1 --shard-mode whole --replicate-mb 64 ;;
f) bash profiles/collect_dist_pmc.sh $round dist-c4-rows-l1 --workload c4 --rows-scale 0.125 --replicate-mb 64 && \
   bash profiles/collect_dist_pmc.sh $round dist-c4-rows-l32 --workload c4 --rows-scale 0.125 --replicate-mb 64 --pooling 32 && \
   DIST_PMC_DIRECT=1 bash profiles/collect_dist_pmc.sh $round dist-c4-rows-l1-direct --workload c4 --rows-scale 0.125 --replicate-mb 64 && \
   DIST_PMC_DIRECT=1 bash profiles/collect_dist_pmc.sh $round dist-c2-rows-l1-direct --shard-mode rows --replicate-mb 64 && \
   DIST_PMC_DIRECT=1 bash profiles/collect_dist_pmc.sh $round dist-c2-whole-l1 --shard-mode whole --replicate-mb 64 ;;
b) STEPS_STATS=60 bash profiles/collect.sh $round c3 --workload c3 && STEPS_STATS=30 bash profiles/collect.sh $round c3-uniform --workload c3 --index-dist uniform ;;
c) STEPS_STATS=60 bash profiles/collect.sh $round c5 --workload c5 ;;
h) STEPS_STATS=60 bash profiles/collect.sh $round c3-hot32 --workload c3 --hot-rows 32 ;;
d) bash tools/run_all_benches.sh gpurun_out/profiles_$round/benches ;;
esac
