# bench.py --gpus 2 rehearsed on ONE GPU: both ranks on device 0, exchange through the host-staged transport (gloo)
mkdir -p gpurun_out/r3g
LSA_BENCH_STACKS=${LSA_BENCH_STACKS:-90} LSA_BENCH_DEVICE=0 LSA_BENCH_BACKEND=gloo timeout -k 10 ${REHEARSE_TIMEOUT:-300} python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r3g/bench_2ranks_rehearsal.json 2> gpurun_out/r3g/bench_2ranks.err
echo "exit $?"
grep -v "^\[W\|amdgpu.ids" gpurun_out/r3g/bench_2ranks.err | tail -60
python3 tools/print_bench_line.py gpurun_out/r3g/bench_2ranks_rehearsal.json
