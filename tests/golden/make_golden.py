"""Generate golden vectors by IMPORTING the reference's importable hot-path pieces.

Run in the build container only (the reference never travels):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

Writes tests/golden/golden_host.json (integer / python-float results) and
tests/golden/golden_tensors.pt (small tensors).  What is imported:

  training_utils/optimizers/raven.py, titan.py   -- import as-is
  train.py helper functions                      -- train.py imports four third-party packages at
     top level that are absent from this image (diffusers, cv2, tomesd, torchvision); inert
     placeholder modules are registered for those names so that the module body executes.  None of
     the captured functions touches them (they are pure torch / numpy / stdlib).

The UNet forward and the DDPM scheduler live in diffusers and cannot be captured (PARITY UNPINNED).
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _placeholder(name, attrs=()):
    m = types.ModuleType(name)
    for a in attrs:
        setattr(m, a, type(a, (), {}))
    sys.modules[name] = m
    return m


def import_reference():
    import importlib
    _placeholder("diffusers", ["StableDiffusionXLPipeline", "DDPMScheduler", "UNet2DConditionModel",
                               "AutoencoderKL", "AutoencoderKLFlux2", "FlowMatchEulerDiscreteScheduler"])
    opt = _placeholder("diffusers.optimization")
    opt.get_scheduler = lambda *a, **k: None
    _placeholder("diffusers.models")
    _placeholder("diffusers.models.attention_processor",
                 ["AttnProcessor2_0", "XFormersAttnProcessor", "AttnProcessor", "FusedAttnProcessor2_0"])
    _placeholder("cv2")
    _placeholder("tomesd")
    tv = _placeholder("torchvision")
    tvt = _placeholder("torchvision.transforms", ["Compose", "ToTensor", "Normalize", "Resize", "InterpolationMode"])
    tv.transforms = tvt
    sys.path.insert(0, REF)
    sys.argv = ["train.py"]

    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return type(k, (), {})
    for n in list(sys.modules):
        if n.startswith("diffusers") or n.startswith("torchvision"):
            m = sys.modules[n]
            any_m = _Any(n)
            any_m.__dict__.update(m.__dict__)
            sys.modules[n] = any_m
    train = importlib.import_module("train")
    raven = importlib.import_module("training_utils.optimizers.raven")
    titan = importlib.import_module("training_utils.optimizers.titan")
    return train, raven, titan


def main():
    import torch
    train, raven, titan = import_reference()
    host = {}
    tens = {}

    # ---- F5 tickets ----------------------------------------------------------------
    logit_counts = [45, 143, 176, 173, 154, 126, 94, 59, 26, 4]
    cases = []
    for alloc, total, seed, strat in [
        (None, 16, 42, False), ({"bin_size": 100, "counts": []}, 64, 42, False),
        ({"bin_size": 100, "counts": logit_counts}, 32000, 42, False),
        ({"bin_size": 100, "counts": logit_counts}, 4096, 7, True),
        ({"bin_size": 50, "counts": list(range(1, 21))}, 1000, 1234, False),
        ({"bin_size": 100, "counts": [0, 0, 5, 0, 5, 0, 0, 0, 0, 1]}, 37, 99, True),
        ({"bin_size": 100, "counts": logit_counts}, 0, 42, False),
        (None, 3, 0, False),
    ]:
        pool, ranges = train.build_timestep_ticket_pool(alloc, total, 1000, seed, strat)
        hist = [sum(1 for t in pool if lo <= t < hi) for lo, hi in ranges]
        cases.append(dict(allocation=alloc, total=total, seed=seed, stratified=strat,
                          head=pool[:64], tail=pool[-8:], length=len(pool), checksum=int(sum((i + 1) * t for i, t in enumerate(pool)) % (2 ** 61 - 1)),
                          ranges=[list(r) for r in ranges], hist=hist))
    host["tickets"] = cases

    class Cfg:
        MAX_TRAIN_STEPS = 10
        BATCH_SIZE = 4
        SEED = 42
        is_rectified_flow = False
        TIMESTEP_ALLOCATION = {"bin_size": 100, "counts": logit_counts}
        TIMESTEP_STRATIFIED_SAMPLING = False
    s = train.TimestepSampler(Cfg, "cpu")
    seq = []
    for _ in range(12):
        t, first = s.sample(4)
        seq.append(t.tolist())
    s.set_current_step(3)
    host["sampler"] = dict(seq=seq, after_set3=s.sample(4)[0].tolist(), state=s.state_dict())

    # ---- F7 lr ------------------------------------------------------------------------
    class Opt:
        def __init__(self):
            self.param_groups = [{"lr": 0.0, "lr_scale": 1.0}, {"lr": 0.0, "lr_scale": 0.5}]
    lr_cases = []
    for curve, total in [([[0.0, 0.0], [0.05, 8.0e-7], [0.85, 8.0e-7], [1.0, 1.0e-7]], 100),
                         ([[0.2, 1e-4], [0.6, 5e-5]], 37), ([[0.0, 1e-5], [1.0, 1e-5]], 1),
                         ([[0.5, 3e-4], [0.5, 1e-4], [1.0, 0.0]], 11)]:
        o = Opt()
        sch = train.CustomCurveLRScheduler(o, [list(p) for p in curve], total)
        vals = []
        for ms in range(0, total + 2):
            sch.step(ms)
            vals.append([g["lr"] for g in o.param_groups])
        lr_cases.append(dict(curve=curve, total=total, lrs=vals))
    host["lr"] = lr_cases

    # ---- F6 rng -------------------------------------------------------------------------
    g = torch.Generator(device="cpu")
    rng_cases = []
    for seed, step, shape in [(42, 1, (1, 4, 2, 2)), (42, 2, (2, 4, 3, 3)), (0, 7, (1, 4, 2, 2)), (2 ** 32 - 5, 9, (1, 4, 2, 2))]:
        n = train.generate_noise(torch.zeros(shape), g, "cpu", step=step, seed=seed)
        key = f"noise_{seed}_{step}"
        tens[key] = n.clone()
        rng_cases.append(dict(seed=seed, step=step, shape=list(shape), key=key))
    jit_cases = []
    for seed, parts, n in [(42, (1, 0x5D1), 4), (42, (2, 0x5D1), 4), (0, (5, 0x5D1), 3), (123456789, (1000, 0x5D1), 8)]:
        gen = train.seeded_torch_generator("cpu", seed, *parts)
        j = torch.rand((n,), dtype=torch.float32, generator=gen)
        key = f"jitter_{seed}_{parts[0]}"
        tens[key] = j.clone()
        jit_cases.append(dict(seed=seed, parts=list(parts), n=n, key=key, initial_seed=int(gen.initial_seed())))
    host["rng"] = dict(noise=rng_cases, jitter=jit_cases)

    # ---- F3/F4 loss + curves -------------------------------------------------------------
    class C1: TIMESTEP_LOSS_WEIGHT_CURVE = [[0.0, 1.0], [1.0, 1.0]]
    class C2: TIMESTEP_LOSS_WEIGHT_CURVE = {"preset": "bell"}
    class C3: TIMESTEP_LOSS_WEIGHT_CURVE = [[0.1, 0.5], [0.5, 2.0], [0.9, 0.25]]
    class C4: TIMESTEP_LOSS_WEIGHT_CURVE = None
    for nm, c in [("flat", C1), ("bell", C2), ("custom", C3), ("none", C4)]:
        tens[f"curve_{nm}"] = train.timestep_loss_curve_from_config(c, 1000)
    torch.manual_seed(0)
    loss_cases = []
    for i, (B, shape, dt) in enumerate([(2, (4, 8, 8), torch.float32), (4, (4, 16, 16), torch.bfloat16), (1, (4, 4, 4), torch.float32)]):
        pred = torch.randn((B,) + shape).to(dt).requires_grad_(True)
        tgt = torch.randn((B,) + shape)
        ts = torch.tensor([10, 900, 499, 0][:B])
        for nm in ["none", "flat", "bell", "custom"]:
            curve = None if nm == "none" else tens[f"curve_{nm}"]
            pred.grad = None
            l = train.weighted_sdxl_mse_loss(pred, tgt, ts, curve)
            l.backward()
            k = f"loss{i}_{nm}"
            tens[k + "_pred"] = pred.detach().clone()
            tens[k + "_tgt"] = tgt.clone()
            tens[k + "_ts"] = ts.clone()
            tens[k + "_loss"] = l.detach().clone()
            tens[k + "_dpred"] = pred.grad.clone()
            loss_cases.append(dict(key=k, curve=nm))
    host["loss"] = loss_cases

    # ---- F1 raven ---------------------------------------------------------------------------
    opt_cases = []
    torch.manual_seed(1)
    ci = 0
    for numel in (1, 7, 4096):
        for pdt in (torch.bfloat16, torch.float32):
            for mdt in (torch.bfloat16, torch.float32):
                for debias in (0.3, 1.0):
                    p0 = (torch.randn(numel) * 0.05).to(pdt)
                    grads = [(torch.randn(numel) * 0.01).to(pdt) for _ in range(3)]
                    p = torch.nn.Parameter(p0.clone())
                    o = raven.RavenAdamW([{"params": [p], "lr_scale": 1.0}], lr=1e-3, betas=(0.9, 0.999),
                                         weight_decay=0.01, eps=1e-8, debias_strength=debias, momentum_dtype=mdt)
                    k = f"raven{ci}"
                    tens[k + "_init"] = p0
                    for s_i, gr in enumerate(grads):
                        p.grad = gr.clone()
                        o.step()
                        tens[f"{k}_g{s_i}"] = gr
                        tens[f"{k}_p{s_i}"] = p.detach().clone()
                        tens[f"{k}_m{s_i}"] = o.state[p]["exp_avg"].clone()
                        tens[f"{k}_v{s_i}"] = o.state[p]["exp_avg_sq"].clone()
                    st = o.save_cpu_state()
                    opt_cases.append(dict(key=k, numel=numel, pdt=str(pdt), mdt=str(mdt), debias=debias, steps=3,
                                          lr=1e-3, betas=[0.9, 0.999], wd=0.01, eps=1e-8,
                                          state_keys=sorted(str(x) for x in st.keys()),
                                          state0_keys=sorted(st[0].keys()), step=int(st[0]["step"])))
                    ci += 1
    host["raven"] = opt_cases

    # ---- F2 titan cycle (GA=2 hook accumulation, CPU clip) --------------------------------------
    titan_cases = []
    torch.manual_seed(2)
    for ti, (max_norm, mdt) in enumerate([(0.5, torch.bfloat16), (1.0, torch.float32), (float("inf"), torch.bfloat16)]):
        w1 = torch.nn.Parameter((torch.randn(8, 5) * 0.3).to(torch.bfloat16))
        w2 = torch.nn.Parameter((torch.randn(3, 8) * 0.3).to(torch.bfloat16))
        o = titan.TitanAdamW([{"params": [w1, w2], "lr_scale": 1.0}], lr=1e-3, betas=(0.9, 0.999),
                             weight_decay=0.01, eps=1e-8, debias_strength=0.3, momentum_dtype=mdt)
        k = f"titan{ti}"
        tens[k + "_w1"] = w1.detach().clone()
        tens[k + "_w2"] = w2.detach().clone()
        xs = [torch.randn(4, 5).to(torch.bfloat16) for _ in range(2)]
        for mi, x in enumerate(xs):
            tens[f"{k}_x{mi}"] = x
            y = (torch.tanh(x @ w1.t()) @ w2.t()).float().pow(2).mean()
            (y / 2).backward()
        grads_none = (w1.grad is None) and (w2.grad is None)
        tens[k + "_cpu_g1"] = o._cpu_grads[w1].clone()
        tens[k + "_cpu_g2"] = o._cpu_grads[w2].clone()
        norm = o.clip_grad_norm(max_norm)
        tens[k + "_norm"] = torch.as_tensor(norm).clone()
        tens[k + "_clip_g1"] = o._cpu_grads[w1].clone()
        tens[k + "_clip_g2"] = o._cpu_grads[w2].clone()
        o.step()
        tens[k + "_w1_after"] = w1.detach().clone()
        tens[k + "_w2_after"] = w2.detach().clone()
        tens[k + "_m1"] = o.state[w1]["exp_avg"].clone()
        tens[k + "_v1"] = o.state[w1]["exp_avg_sq"].clone()
        o.zero_grad(set_to_none=True)
        ready_after = len(o._cpu_grad_ready)
        titan_cases.append(dict(key=k, max_norm=("inf" if max_norm == float("inf") else max_norm), mdt=str(mdt),
                                grads_none_after_backward=grads_none, ready_after_zero_grad=ready_after))
        o.close()
    host["titan"] = titan_cases
    # double-ownership error behaviour
    w = torch.nn.Parameter(torch.zeros(3))
    o1 = titan.TitanAdamW([w])
    try:
        titan.TitanAdamW([w])
        host["titan_double_owner"] = "no error"
    except RuntimeError as e:
        host["titan_double_owner"] = "RuntimeError"
    o1.close()
    for bad in ("lr", "dtype"):
        try:
            if bad == "lr":
                raven.RavenAdamW([torch.nn.Parameter(torch.zeros(1))], lr=-1.0)
            else:
                raven.RavenAdamW([torch.nn.Parameter(torch.zeros(1))], momentum_dtype=torch.float64)
            host[f"raven_bad_{bad}"] = "no error"
        except ValueError:
            host[f"raven_bad_{bad}"] = "ValueError"

    # ---- F11 clip -------------------------------------------------------------------------------
    torch.manual_seed(3)
    for ci2, dt in enumerate([torch.bfloat16, torch.float32]):
        ps = [torch.nn.Parameter(torch.zeros(n).to(dt)) for n in (5, 64, 1000)]
        for p in ps:
            p.grad = (torch.randn(p.shape) * 0.2).to(dt)
        for i, p in enumerate(ps):
            tens[f"clip{ci2}_g{i}"] = p.grad.clone()
        n = torch.nn.utils.clip_grad_norm_(ps, 1.0)
        tens[f"clip{ci2}_norm"] = n.clone()
        for i, p in enumerate(ps):
            tens[f"clip{ci2}_c{i}"] = p.grad.clone()

    # ---- F12 key map + F10 freeze -------------------------------------------------------------------
    sys.path.insert(0, "/root/repo")
    from oracle.unet_ref import param_table, SDXL_BASE
    names = [n for n, _ in param_table(SDXL_BASE)]
    mapping = train.get_unet_key_mapping(names)
    import hashlib
    sd_names = [mapping[n] for n in names]
    host["keymap"] = dict(n=len(mapping), unique_targets=len(set(sd_names)),
                          digest=hashlib.sha256("\n".join(f"{a}->{mapping[a]}" for a in names).encode()).hexdigest(),
                          samples={n: mapping[n] for n in names[::97]})
    import fnmatch
    shapes = dict(param_table(SDXL_BASE))
    import math as _m
    fz = []
    for kws in (["conv1", "conv2"], ["mid_block", "up_blocks.3"], ["attn2*to_k*"], []):
        frozen = [n for n in names if any(fnmatch.fnmatch(n, kw if "*" in kw else f"*{kw}*") for kw in kws)]
        fz.append(dict(keywords=kws, n_frozen=len(frozen), frozen_numel=sum(_m.prod(shapes[n]) for n in frozen),
                       first=frozen[:3], last=frozen[-3:]))
    host["freeze"] = fz

    with open(os.path.join(OUT, "golden_host.json"), "w") as f:
        json.dump(host, f, indent=0, sort_keys=True)
    torch.save(tens, os.path.join(OUT, "golden_tensors.pt"))
    print("wrote", len(host), "host groups,", len(tens), "tensors")


if __name__ == "__main__":
    main()
