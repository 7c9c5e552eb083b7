import os, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world):
    os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]="29544"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev=torch.device("cuda",0)
    for name, fn in (("all_reduce", lambda: dist.all_reduce(torch.ones(4,device=dev))),
                     ("a2a", lambda: dist.all_to_all_single(torch.empty(4,device=dev), torch.ones(4,device=dev))),
                     ("agit", lambda: dist.all_gather_into_tensor(torch.empty(8,device=dev), torch.ones(4,device=dev))),
                     ("barrier", lambda: dist.barrier())):
        try:
            fn(); torch.cuda.synchronize(); print(rank, name, "ok")
        except Exception as e:
            print(rank, name, "FAIL", repr(e)[:150])
    dist.destroy_process_group()
if __name__=="__main__":
    mp.spawn(w,args=(2,),nprocs=2,join=True)
