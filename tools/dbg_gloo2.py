
import os, sys, json, torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.environ["CN_REPO"])
from centernet_amd import rng, synth
from centernet_amd.engine import TrainStep
from centernet_amd.centernet_detection import CenterNetDetection

def make(arch):
    m = CenterNetDetection(arch, compute_dtype=torch.float32)
    rng.fill_state_dict(m, 97)
    return m.cuda().train()

def batch_of(rank):
    x, tgt = synth.ctdet_batch(97, 2, 128, 128, start=2 * rank)
    return x.cuda(), {k: v.cuda() for k, v in tgt.items()}

def worker(rank, world, port, arch, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)          # two ranks on ONE GPU: RCCL refuses that, gloo does not
    step = TrainStep(make(arch), lr=0.0, graph=False)
    assert step.sync is not None and step.sync.exchange and step.side
    b = batch_of(rank)
    step(b)                                   # learns which parameters are live; every bucket leaves in finish()
    step(b)                                   # buckets leave while backward is still producing gradients
    torch.cuda.synchronize()
    out[rank] = (step.opt.flat_g.cpu(), list(step.sync.launch_log), len(step.sync.live))
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    arch = sys.argv[1]
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(worker, args=(2, int(os.environ["CN_PORT"]), arch, out), nprocs=2, join=True)
    ref = 0
    refs = []
    for r in (0, 1, 0, 1):                         # what each rank computes on its own, no exchange
        step = TrainStep(make(arch), lr=0.0, distributed=False, graph=False)
        step(batch_of(r)); step(batch_of(r))
        torch.cuda.synchronize()
        refs.append(step.opt.flat_g.cpu())
    ref = refs[0] + refs[1]
    n = 64 * 3 * 49
    a = out[0][0][:n]
    print("STEM got", float(a.abs().max()), "r0", float(refs[0][:n].abs().max()), "r1", float(refs[1][:n].abs().max()),
          "r0 again diff", float((refs[0][:n] - refs[2][:n]).abs().max()), "r1 again diff", float((refs[1][:n] - refs[3][:n]).abs().max()),
          "got-r0", float((a - refs[0][:n]).abs().max()), "got-r1", float((a - refs[1][:n]).abs().max()),
          "got-sum", float((a - refs[0][:n] - refs[1][:n]).abs().max()), "rank1 got - rank0 got", float((out[1][0][:n] - a).abs().max()))
    g0, log, live = out[0]
    g1 = out[1][0]
    err = float((g0 - ref).abs().max() / ref.abs().max())
    m = make(arch); names = [n for n, p in m.named_parameters() if p.requires_grad]
    st = TrainStep(m, lr=0.0, distributed=False, graph=False)
    worst = []
    for n, p, o in zip(names, st.opt.params, st.opt.offsets):
        a, b = g0[o:o + p.numel()], ref[o:o + p.numel()]
        e = float((a - b).abs().max() / (b.abs().max() + 1e-12))
        worst.append((e, n, float(b.abs().max()), float(a.abs().max())))
    worst.sort(reverse=True)
    for w in worst[:25]: print("PARAM", w)
    print("n bad", sum(1 for w in worst if w[0] > 1e-3), "of", len(worst))
    print("RESULT " + json.dumps({"same": bool(torch.equal(g0, g1)), "err": err, "log": log, "live": live,
                                  "nonzero": float(ref.abs().max())}))
