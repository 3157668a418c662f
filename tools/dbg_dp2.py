import os, sys, types
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, torch.multiprocessing as mp
import test_dp_gpu as T

def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0); dev = "cuda:0"
    from aozora_sdxl_training_amd.unet import AozoraUNet
    from aozora_sdxl_training_amd.unet_spec import mini_config
    from aozora_sdxl_training_amd.train_step import TrainStep
    from aozora_sdxl_training_amd.dist import ShardedRaven
    pc = mini_config(); g = torch.Generator().manual_seed(1234)
    def make_unet():
        u = AozoraUNet(pc, dev); gg = torch.Generator().manual_seed(77)
        with torch.no_grad():
            for n, p in u.named_parameters():
                if "norm" in n: p.fill_(1.0 if n.endswith("weight") else 0.0)
                else: p.copy_((torch.randn(p.shape, generator=gg) * 0.05).bfloat16())
        return u
    GB, h, w = 4, 16, 16
    lat = torch.randn(GB, 4, h, w, generator=g).bfloat16(); noise = torch.randn(GB, 4, h, w, generator=g)
    ctx = torch.randn(GB, 77, pc.cross_attention_dim, generator=g).bfloat16(); pooled = torch.randn(GB, pc.pooled_dim, generator=g).bfloat16()
    tid = torch.tensor([[128, 128, 0, 0, 128, 128]] * GB, dtype=torch.bfloat16)
    ts = torch.tensor([37, 911, 500, 250]); b = GB // world; sl = slice(rank * b, (rank + 1) * b)
    GA = 2
    runs = {}
    for tag, kw, hook in (("A", dict(), os.environ.get("DBG_HOOK", "1") == "1"), ("B", dict(overlap=False, regions=3), False)):
        u = make_unet(); step = TrainStep(u, mode="epsilon", grad_accum=GA, world_size=world, use_graph=False)
        opt = ShardedRaven(u, lr=1e-4, clip_grad_norm=1.0, **kw)
        snaps = []
        for it in range(2):
            u.zero_grad()
            for m in range(GA):
                k = it * GA + m
                args = [t.roll(k, 0)[sl].to(dev) for t in (lat, noise)] + [ts.roll(k, 0)[sl]] + [t.roll(k, 0)[sl].to(dev) for t in (ctx, pooled, tid)]
                p_before = u.pflat.clone()
                lossv = step.micro_step(*args, after_tail=(opt.reduce_tail if (hook and m == GA - 1) else None))
                torch.cuda.synchronize()
                snaps.append(("x", it, m, (float(lossv.item()), u.wtflat.clone(), p_before, step.last_pred_nhwc.clone())))
                if not (hook and m == GA - 1):
                    snaps.append(("g", it, m, u.gflat.clone()))
            gn = opt.step().item(); u.wait_tail_params(); torch.cuda.synchronize()
            snaps.append(("p", it, gn, u.pflat.clone()))
        runs[tag] = (u, snaps)
    ua, sa = runs["A"]; ub, sb = runs["B"]
    names = [(n, o, n_) for n, (o, s_, n_) in ua._slots.items()] if False else None
    for (ka, ita, xa, ta), (kb, itb, xb, tb) in zip([s for s in sa if s[0] == "p"], [s for s in sb if s[0] == "p"]):
        diff = (ta != tb)
        print(f"[rank {rank}] after opt step {ita}: gn A {xa} B {xb}; params differing: {int(diff.sum())}", flush=True)
        if diff.any():
            idx = diff.nonzero().flatten()
            regs = ua.region_bounds()
            print(f"[rank {rank}]   first diff offset {int(idx[0])}, last {int(idx[-1])}; regions {regs}; own A {None}", flush=True)
    ga = [s for s in sa if s[0] == "g"]; gb = [s for s in sb if s[0] == "g"]
    xa = {(it_, m_): v for (k_, it_, m_, v) in sa if k_ == "x"}; xb = {(it_, m_): v for (k_, it_, m_, v) in sb if k_ == "x"}
    for key in sorted(xa):
        la, wa, pa, pra = xa[key]; lb, wb, pb, prb = xb[key]
        print(f"[rank {rank}] micro {key}: loss A {la} B {lb}; wtflat differing {int((wa != wb).sum())}; pflat-before differing {int((pa != pb).sum())}; pred differing {int((pra != prb).sum())}", flush=True)
    db = {(it2, m2): t2 for (k2, it2, m2, t2) in gb}
    for (k1, it1, m1, t1) in ga:
        t2 = db[(it1, m1)]
        d = (t1 != t2)
        print(f"[rank {rank}] grads after micro-step it{it1} m{m1}: differing {int(d.sum())}" + (f" first {int(d.nonzero()[0])} last {int(d.nonzero()[-1])}" if d.any() else ""), flush=True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == '__main__':
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(worker, args=(2, T._free_port(), out), nprocs=2, join=True)
