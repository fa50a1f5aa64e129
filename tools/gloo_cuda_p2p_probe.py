import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev="cuda"
    for trial in range(3):
        n=4_000_000
        if rank==0:
            a=torch.randn(n, device=dev, dtype=torch.float64)
            for _ in range(20): a = a*1.0000001+1e-9   # queued kernels
            t=a
            reqs=[dist.P2POp(dist.isend, t, 1)]
        else:
            t=torch.zeros(n, device=dev, dtype=torch.float64)
            reqs=[dist.P2POp(dist.irecv, t, 0)]
        for wk in dist.batch_isend_irecv(reqs): wk.wait()
        s=t.sum().item()
        lst=[None,None]; dist.all_gather_object(lst, s)
        if rank==0: print("trial",trial,"sums",lst, "equal", lst[0]==lst[1], flush=True)
    dist.destroy_process_group()
if __name__=="__main__":
    import socket
    s=socket.socket(); s.bind(("127.0.0.1",0)); port=s.getsockname()[1]; s.close()
    mp.spawn(w, args=(2,port), nprocs=2, join=True)
