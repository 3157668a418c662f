"""Per-shape timing of the normalisation kernels at the SDXL micro-step's shapes (B=4, 1024x1024): HIP events around `reps`
back-to-back calls on one stream, operands rotated through enough buffers that nothing stays in the Infinity Cache.
python3 tools/norm_bench.py [gn|ln|all] [reps]  -> us per call, effective GB/s (algorithmic bytes), total ms per micro-step."""
import sys, torch
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import ops

dev = torch.device('cuda', 0)
BF, F32 = torch.bfloat16, torch.float32
what = sys.argv[1] if len(sys.argv) > 1 else 'all'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = 4
# (HW, C, silu, calls per micro-step)
GN = [(16384, 320, 1, 5), (16384, 640, 1, 2), (16384, 960, 1, 1), (4096, 320, 1, 1), (4096, 640, 1, 6), (4096, 640, 0, 5), (4096, 960, 1, 1),
      (4096, 1280, 1, 1), (4096, 1920, 1, 1), (1024, 640, 1, 1), (1024, 1280, 1, 12), (1024, 1280, 0, 6), (1024, 1920, 1, 1), (1024, 2560, 1, 2)]
LN = [(4096, 1280, 180), (16384, 640, 30)]


def timeit(fn, n):
    fn(0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def rot(shape, dtype=BF, n=None):
    nbytes = 2
    for s in shape: nbytes *= s
    k = n or max(2, min(12, (600 << 20) // nbytes))
    return [torch.randn(shape, device=dev, dtype=F32).to(dtype) for _ in range(k)]


tot = {}
if what in ('gn', 'all'):
    for HW, C, silu, calls in GN:
        xs, ys, dys, dxs, adds = rot((B, HW, C)), rot((B, HW, C)), rot((B, HW, C)), rot((B, HW, C)), rot((B, HW, C))
        k = len(xs)
        g, b = torch.randn(C, device=dev).to(BF), torch.randn(C, device=dev).to(BF)
        dg, db = torch.zeros(C, device=dev, dtype=BF), torch.zeros(C, device=dev, dtype=BF)
        st = torch.empty(B * 32 * 2, device=dev, dtype=F32)
        tf = timeit(lambda i: ops.groupnorm_fwd(xs[i % k], g, b, ys[i % k], st, 32, 1e-5, silu), reps)
        tb = timeit(lambda i: ops.groupnorm_bwd(xs[i % k], g, b, st, dys[i % k], dxs[i % k], dg, db, 32, silu, dx_add=adds[i % k]), reps)
        n = B * HW * C * 2
        print(f"gn  {HW:6d}x{C:5d} silu={silu} x{calls:3d}: fwd {tf:7.1f} us ({3 * n / tf / 1e3:6.0f} GB/s)   bwd {tb:7.1f} us ({6 * n / tb / 1e3:6.0f} GB/s)", flush=True)
        tot['gn_fwd'] = tot.get('gn_fwd', 0) + tf * calls; tot['gn_bwd'] = tot.get('gn_bwd', 0) + tb * calls
if what in ('ln', 'all'):
    for M, C, calls in LN:
        xs, ys, dys, dxs, adds = rot((M, C)), rot((M, C)), rot((M, C)), rot((M, C)), rot((M, C))
        k = len(xs)
        g, b = torch.randn(C, device=dev).to(BF), torch.randn(C, device=dev).to(BF)
        st = torch.empty(M * 2, device=dev, dtype=F32)
        part = torch.empty(ops.ln_partial_blocks(M) * C * 2, device=dev, dtype=F32)
        tf = timeit(lambda i: ops.layernorm_fwd(xs[i % k], g, b, ys[i % k], st), reps)
        tb = timeit(lambda i: ops.layernorm_bwd_partial(xs[i % k], g, st, dys[i % k], dxs[i % k], part, dx_add=adds[i % k]), reps)
        n = M * C * 2
        print(f"ln  {M:6d}x{C:5d}        x{calls:3d}: fwd {tf:7.1f} us ({2 * n / tf / 1e3:6.0f} GB/s)   bwd {tb:7.1f} us ({4 * n / tb / 1e3:6.0f} GB/s)", flush=True)
        tot['ln_fwd'] = tot.get('ln_fwd', 0) + tf * calls; tot['ln_bwd'] = tot.get('ln_bwd', 0) + tb * calls
print("per micro-step, ms:", {k: round(v / 1e3, 2) for k, v in tot.items()}, flush=True)
