import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.full((4, 3), float(rank), device="cuda", dtype=torch.float64)
    out = torch.empty((world * 4, 3), device="cuda", dtype=torch.float64)
    try:
        work = dist.all_gather_into_tensor(out, x, async_op=True); work.wait(); torch.cuda.synchronize()
        print(rank, "ok", out[:, 0].tolist(), flush=True)
        # in place
        out2 = torch.zeros((world * 4, 3), device="cuda", dtype=torch.float64); out2[rank*4:(rank+1)*4] = rank + 10
        dist.all_gather_into_tensor(out2, out2[rank*4:(rank+1)*4]); torch.cuda.synchronize()
        print(rank, "in-place", out2[:, 0].tolist(), flush=True)
    except Exception as e:
        print(rank, "FAILED", repr(e)[:300], flush=True)
    dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(w, args=(2, 29777), nprocs=2, join=True)
