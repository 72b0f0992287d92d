cd $GRAFT_REPO_ROOT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
python - <<'PY'
import time
t0=time.time(); import torch; print("import torch %.1f s" % (time.time()-t0), flush=True)
t0=time.time(); import torch.distributed as dist; torch.cuda.set_device(0); print("set_device %.1f s" % (time.time()-t0), flush=True)
t0=time.time(); dist.init_process_group("nccl", device_id=torch.device("cuda",0)); print("init_process_group %.1f s" % (time.time()-t0), flush=True)
t0=time.time(); x=torch.ones(4,device="cuda"); dist.all_reduce(x); torch.cuda.synchronize(); print("first all_reduce %.1f s" % (time.time()-t0), flush=True)
t0=time.time(); a=torch.arange(8,dtype=torch.int32,device="cuda"); b=torch.empty_like(a); dist.all_to_all_single(b,a); torch.cuda.synchronize(); print("first all_to_all %.1f s" % (time.time()-t0), flush=True)
dist.destroy_process_group()
PY
