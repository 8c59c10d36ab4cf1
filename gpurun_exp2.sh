cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --cpu-frames 0 --shard-map > gpurun_out/o1.txt 2> gpurun_out/e1.txt; tail -1 gpurun_out/o1.txt | cut -c1-80; wc -l gpurun_out/o1.txt
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 30 --warmup 5 > gpurun_out/o2.txt 2> gpurun_out/e2.txt; tail -1 gpurun_out/o2.txt | cut -c1-80; wc -l gpurun_out/o2.txt; head -3 gpurun_out/o2.txt | cut -c1-80
timeout 300 python bench.py > gpurun_out/o3.txt 2> gpurun_out/e3.txt; tail -1 gpurun_out/o3.txt | cut -c1-120; wc -l gpurun_out/o3.txt
