import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for log2n in (26, 27, 28):
    n = 1 << log2n
    send = torch.arange(n, dtype=torch.int64, device=dev) * 3 + 1
    recv = torch.zeros(n, dtype=torch.int64, device=dev)
    w = dist.all_to_all_single(recv, send, output_split_sizes=[n], input_split_sizes=[n], async_op=True)
    w.wait(); torch.cuda.synchronize()
    bad = int((recv != send).sum())
    first = int((recv != send).nonzero()[0]) if bad else -1
    print("int64 elements 2^%d (%d MiB): mismatches %d first %d" % (log2n, n * 8 >> 20, bad, first), flush=True)
dist.destroy_process_group()
