import os, torch, torch.distributed as td
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
td.init_process_group("nccl", device_id=dev)
t = torch.arange(8, dtype=torch.float32, device=dev)
td.all_reduce(t, op=td.ReduceOp.AVG); print("avg ok", t.tolist())
td.barrier(device_ids=[0]); print("barrier ok")
x = torch.tensor([1.5], device=dev, dtype=torch.float64); td.all_reduce(x, op=td.ReduceOp.MAX); print("max ok", x.item())
o = torch.randperm(5).to(dev); td.broadcast(o, 0); print("bcast ok")
# capture test: all_reduce inside a graph
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    td.all_reduce(t, op=td.ReduceOp.AVG)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        td.all_reduce(t, op=td.ReduceOp.AVG)
    g.replay(); torch.cuda.synchronize(); print("capture ok")
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:200])
td.destroy_process_group()
