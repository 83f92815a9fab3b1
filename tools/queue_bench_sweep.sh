# The reference's serving shapes on host pointers, client threads against one front end: one by one vs the request queue.
B=pim-embedding-lookup_amd/lib/emb_queue_bench
for c in 1 4 8; do timeout -k 5 120 $B 26 16 100000 1 $c 2000 1 | tail -4; done
for w in 4 16 64; do timeout -k 5 120 $B 26 16 100000 1 4 4096 $w | tail -4; done
timeout -k 5 120 $B 26 16 100000 32 4 2048 16 | tail -4
