"""Host-side cost of the per-fluid-step communication calls of strips.py (one-rank RCCL group, self-send):
how long does the Python/torch.distributed/RCCL launch path take per step, i.e. can it keep up with a ~0.3 ms
GPU step at 8 GPUs?"""
import os, sys, time, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
strips = ge.load_package().strips_module()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29641", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
comm = strips.TorchComm(dist)
H = 11 * 4096 * 9
send = [torch.zeros(H, dtype=torch.float64, device="cuda") for _ in range(2)]
recv = [torch.zeros(H, dtype=torch.float64, device="cuda") for _ in range(2)]
fhf = torch.zeros(150000, dtype=torch.float64, device="cuda").view(torch.int64)
ops = [(0, send[0], recv[0]), (0, send[1], recv[1])]
def step():
    p = comm.exchange_begin(ops)
    comm.exchange_end(p)
    comm.all_reduce_bits(fhf)
for _ in range(20): step()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()          # host time to ENQUEUE (the GPU may lag behind)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e6*(t1-t0)/n:.1f} us per step; incl. GPU completion {1e6*(t2-t0)/n:.1f} us per step")
t0 = time.perf_counter()
for _ in range(n):
    p = comm.exchange_begin(ops); comm.exchange_end(p)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"halo exchange only: host {1e6*(t1-t0)/n:.1f} us, total {1e6*(t2-t0)/n:.1f} us")
t0 = time.perf_counter()
for _ in range(n): comm.all_reduce_bits(fhf)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"all-reduce only: host {1e6*(t1-t0)/n:.1f} us, total {1e6*(t2-t0)/n:.1f} us")
dist.destroy_process_group()
