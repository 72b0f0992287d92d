# 2-rank dry run of bench.py's default (weak) N > 1 flow on a one-GPU box: ranks share the GPU, the exchange goes over gloo
cd $GRAFT_REPO_ROOT
export PANTAX_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29546 bench.py --gpus 2 --steps 6 --warmup 3 --workload custom --species 16 --reads 1600000 --genome-len 1000000 > gpurun_out/weak2.json 2> gpurun_out/weak2.err
echo rc=$?; grep -v "^\[W\|amdgpu.ids\|OMP_NUM\|\*\*\*\*" gpurun_out/weak2.err | tail -5
python3 -c "
import json; d=[json.loads(l) for l in open('gpurun_out/weak2.json') if l.startswith('{')][0]; print(d['value'], d['ms_per_step'], d['n_gpus'], d['scaling'], d['config']['exchange'], d['result']['n_species_rows'], d['result']['n_strain_rows'])"
