import os, sys
if len(sys.argv) > 1: os.environ["TORCH_NCCL_TRACE_BUFFER_SIZE"] = sys.argv[1]; os.environ["TORCH_FR_BUFFER_SIZE"] = sys.argv[1]
import pickle, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29511")
dist.init_process_group("nccl", rank=0, world_size=1)
from torch._C import _distributed_c10d as c10d
x = torch.ones(1<<20, device="cuda")
def dump(tag):
    d = pickle.loads(c10d._dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False))
    ent = d.get("entries", [])
    print(tag, "pg_status", d.get("pg_status"), "entries", len(ent), [(e.get("state"), e.get("retired"), e.get("collective_seq_id")) for e in ent][-4:], flush=True)
dump("before")
w = [dist.all_reduce(x, async_op=True) for _ in range(3)]
dump("after issue")
torch.cuda.synchronize()
dump("after sync")
for i in range(6):
    time.sleep(0.03); dump(f"+{30*(i+1)}ms")
dist.destroy_process_group()
