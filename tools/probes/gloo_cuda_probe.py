import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world, port):
    os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]=str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    x = torch.full((4,), float(rank+1), device="cuda")
    dist.all_reduce(x)
    y = torch.full((2,3), float(rank), device="cuda")
    dist.broadcast(y, src=0)
    print(rank, x.tolist(), y.sum().item(), flush=True)
    dist.destroy_process_group()
if __name__ == "__main__":
    mp.get_context("spawn")
    mp.spawn(w, args=(2, 29511), nprocs=2)
