"""Does an async RCCL all_reduce issued behind a busy stream block the HOST?  (one rank; python tools/probe/nccl_block_probe.py)"""
import os, socket, time, torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
g = torch.randn(8 << 20, device=dev)
a = torch.randn(8192, 8192, device=dev)
side = torch.cuda.Stream()
dist.all_reduce(g)                      # communicator warm-up
torch.cuda.synchronize()

def busy(stream, n):
    with torch.cuda.stream(stream):
        x = a
        for _ in range(n):
            x = x @ a
    return x

for label, ctx in (("side-stream context", side), ("main-stream context", torch.cuda.current_stream())):
    for n in (0, 20):
        torch.cuda.synchronize()
        busy(ctx, n)                     # ~n x 1 ms of queued work on the stream the collective is ordered behind
        t0 = time.perf_counter()
        with torch.cuda.stream(ctx):
            w = dist.all_reduce(g, async_op=True)
        t1 = time.perf_counter()
        w.wait()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print("%s, %2d queued GEMMs: all_reduce call %.3f ms on the host, wait() %.3f ms, drain %.3f ms" % (label, n, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
dist.destroy_process_group()
