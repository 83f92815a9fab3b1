bash profiles/collect_dist_pmc.sh r04 dist-c4-rows-l1 --workload c4 --rows-scale 0.125 --replicate-mb 64 && \
bash profiles/collect_dist_pmc.sh r04 dist-c4-rows-l32 --workload c4 --rows-scale 0.125 --replicate-mb 64 --pooling 32
