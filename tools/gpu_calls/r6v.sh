# the driver's multi-GPU launch form with one rank (RCCL initialised, world size 1) on the final tree
O=gpurun_out/r6v; mkdir -p $O
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 > $O/bench_torchrun_world1.log 2>&1 || { tail -20 $O/bench_torchrun_world1.log; exit 1; }
tail -1 $O/bench_torchrun_world1.log | cut -c1-300
