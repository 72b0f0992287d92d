cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_route.py -m gpu -x -q -k rccl 2>&1 | tail -15
export PANTAX_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 5 --warmup 2 --scaling strong --workload custom --species 12 --reads 1200000 --genome-len 1000000 > gpurun_out/r02m_strong2.json 2> gpurun_out/r02m_strong2.err
echo rc=$?; tail -3 gpurun_out/r02m_strong2.err; python3 tools/bench_summary.py gpurun_out/r02m_strong2.json | head -3
python3 -c "
import json; d=[json.loads(l) for l in open('gpurun_out/r02m_strong2.json') if l.startswith('{')][0]; print(d['ingest_route']); print(d['config']); print(d['result'])"
unset PANTAX_BENCH_BACKEND
timeout 600 python bench.py --steps 5 --warmup 2 --workload custom --species 12 --reads 1200000 --genome-len 1000000 --no-cpu-baseline --no-hard --no-gaf > gpurun_out/r02m_one.json 2> gpurun_out/r02m_one.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r02m_one.json').read().strip().splitlines()[0]); print(d['value'], d['result'])"
