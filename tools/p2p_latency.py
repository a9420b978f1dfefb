"""Latency of one lili_p2p exchange (through gpurun): `world` processes sharing GPU 0, each enqueues n all-reduces of the 72-double record back to
back on its stream; wall time per exchange = launch + store + flag round + rank-order sum (the link is HBM here, not xGMI)."""
import os, sys, socket, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch, torch.distributed as dist
    import lili_om_amd as L
    from lili_om_amd import p2p
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    ctx = L.Context(0, st.cuda_stream)
    comm = p2p.Communicator(ctx, rank, world, dist)
    buf = torch.full((72,), float(rank + 1), dtype=torch.float64, device="cuda")
    stream = st.cuda_stream
    for n in (200, 2000):
        dist.barrier(); ctx.sync(); torch.cuda.synchronize()
        tic = time.perf_counter()
        for _ in range(n):
            comm.all_reduce(buf.data_ptr(), 72, 8, stream)
        ctx.sync(); torch.cuda.synchronize()
        el = time.perf_counter() - tic
        buf.fill_(float(rank + 1)); torch.cuda.synchronize()
    out[rank] = (el / n * 1e6, comm.status())
    dist.barrier(); comm.close(); ctx.close(); dist.destroy_process_group()

if __name__ == "__main__":
    import torch.multiprocessing as mp
    for world in (2, 4):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mgr = mp.Manager(); out = mgr.dict()
        mp.spawn(worker, args=(world, port, out), nprocs=world, join=True)
        print(f"world {world}: " + ", ".join(f"rank {r}: {out[r][0]:.2f} us per exchange (status {out[r][1]})" for r in sorted(out.keys())))
