#!/bin/bash
# rehearsal of the N = 2 bench path on a one-GPU box: both ranks share cuda:0 and exchange over gloo (the number it
# prints is not a scaling measurement)
cd $GRAFT_REPO_ROOT
CAFE_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --repetitions 3 --preheat 50 --config cfg2
