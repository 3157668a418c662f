"""Which pool streams really run beside a given side stream?  For the first 8 high-priority pool streams: (a) single-spin
probe, (b) chained-spin probe (3 dependent kernels per stream), (c) a proxy workload of 300 dependent mid-size GEMMs on each
stream, alone and side by side."""
import ctypes, sys, time, torch
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import ops
from aozora_sdxl_training_amd._lib import lib
dev = 'cuda:0'
L = lib()
P = lambda s: ctypes.c_void_p(s.cuda_stream)
a = torch.randn(1024, 1280, device=dev).bfloat16(); w = torch.randn(1280, 1280, device=dev).bfloat16()
c1 = torch.empty(1024, 1280, device=dev, dtype=torch.bfloat16); c2 = torch.empty_like(c1)
side = torch.cuda.Stream(device=dev)
def spin(streams, n, us):
    for s in streams: s.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for s in streams: L.call("az_spin", us, P(s))
    for s in streams: s.synchronize()
    return (time.perf_counter() - t0) * 1e6
def gemms(streams, outs, n=300):
    for s in streams: s.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for s, o in zip(streams, outs):
            with torch.cuda.stream(s): ops.gemm(a, w, o, trans_b=True)
    for s in streams: s.synchronize()
    return (time.perf_counter() - t0) * 1e3
def pingpong(x, y, rounds=20, us=30):
    """x and y alternate through events: x spins, y waits for it and spins, x waits for y ... (the executor's fork / join pattern)"""
    x.synchronize(); y.synchronize()
    t0 = time.perf_counter()
    for _ in range(rounds):
        L.call("az_spin", us, P(x)); e = torch.cuda.Event(); e.record(x); y.wait_event(e)
        L.call("az_spin", us, P(y)); e2 = torch.cuda.Event(); e2.record(y); x.wait_event(e2)
    x.synchronize(); y.synchronize()
    return (time.perf_counter() - t0) * 1e6
def forkjoin(x, y, rounds=20, us=30):
    """x forks work to y every round and joins it two rounds later, both keep running (wgrad branch pattern)"""
    x.synchronize(); y.synchronize()
    t0 = time.perf_counter(); pend = []
    for r in range(rounds):
        L.call("az_spin", us, P(x)); e = torch.cuda.Event(); e.record(x); y.wait_event(e)
        L.call("az_spin", 2 * us, P(y)); e2 = torch.cuda.Event(); e2.record(y); pend.append(e2)
        L.call("az_spin", us, P(x))
        if len(pend) > 2: x.wait_event(pend.pop(0))
    x.synchronize(); y.synchronize()
    return (time.perf_counter() - t0) * 1e6
spin([side], 1, 10); gemms([side], [c1], 20)
print(f"side alone: 300 GEMMs {gemms([side], [c1]):.2f} ms")
for k in range(8):
    hp = torch.cuda.Stream(device=dev, priority=-1)
    gemms([hp], [c2], 20)
    print(f"hp{k} (0x{hp.cuda_stream:x}): single spin pair {spin([hp, side], 1, 400):6.0f} us | chained x3 {spin([hp, side], 3, 150):6.0f} us "
          f"(alone {spin([hp], 3, 150):6.0f}) | ping-pong 20x2x30us {pingpong(hp, side):6.0f} {pingpong(hp, side):6.0f} us | fork/join {forkjoin(hp, side):6.0f} {forkjoin(hp, side):6.0f} us", flush=True)
