"""Timing rehearsal of ONE rank of an N-rank data-parallel run on a one-GPU box: the shard arithmetic of ShardedRaven sees world = N
(rank 0), so the update, the m / v host-link traffic and the collectives' byte counts are a rank's 1/N share, and the iteration has
32 / (4 N) micro-steps as in bench.py; the collectives themselves run over RCCL in the real group of one rank (no xGMI time: what is
rehearsed is everything else -- host issue rate, stream choreography, copies, the boundary).  The parameters it produces are
meaningless (only rank 0's shard is ever updated).   PRETEND_WORLD=8 python tools/rank_rehearsal.py [iterations]"""
import os, socket, sys, time, torch
sys.path.insert(0, '.')
import bench
import torch.distributed as dist
from aozora_sdxl_training_amd import streams as _streams
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
from aozora_sdxl_training_amd.dist import ShardedRaven
N = int(os.environ.get('PRETEND_WORLD', '8'))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
_streams.host_link_streams(dev)
sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
dist.init_process_group(backend='nccl', device_id=dev)
unet = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(unet)
ga = max(1, 32 // (4 * N))
step = TrainStep(unet, mode='epsilon', grad_accum=ga, world_size=N, use_graph=False)
# the pretence: world size / rank as the shard arithmetic sees them; collectives shrink to the owned slice (valid in a group of one)
real_ws, real_rs, real_ag = dist.get_world_size, dist.reduce_scatter_tensor, dist.all_gather_into_tensor
dist.get_world_size = lambda group=None: N
dist.reduce_scatter_tensor = lambda out, inp, op=None, group=None: real_rs(out, out, op=op or dist.ReduceOp.SUM, group=group)
dist.all_gather_into_tensor = lambda out, inp, group=None: real_ag(inp, inp, group=group)
opt = ShardedRaven(unet, lr=8e-7, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, debias_strength=0.3, momentum_dtype=torch.bfloat16, clip_grad_norm=1.0)
assert opt.world == N and opt.exchange and opt.overlap
if os.environ.get('COMM_ON') == 'h2d': opt.comm = opt.copy_streams[0]        # experiments: the exchange stream folded into another one
if os.environ.get('COMM_ON') == 'd2h': opt.comm = opt.copy_streams[1]
if os.environ.get('COMM_ON') == 'side': opt.comm = unet._sides[0]
if os.environ.get('COMM_ON') == 'pick': opt.comm = _streams.pick(dev, beside=[step.stream, unet._sides[0]], what='exchange stream')
batch = bench.synthetic_batch(0, 0, 0, 4, dev)
def iteration():
    for m in range(ga):
        if m == max(0, ga - bench.PREFETCH_LEAD): opt.prefetch()
        step.micro_step(*batch, after_tail=opt.reduce_tail if m == ga - 1 else None)
    opt.step()
    if os.environ.get('CLEAR_ON_MAIN') == '1': opt._upd_ev = None       # A/B: the gradient clear on the main stream (the form before round 4's change)
    opt.zero_grad(set_to_none=True)
for _ in range(3): step.micro_step(*batch)
step.synchronize(); opt.zero_grad(set_to_none=True)
for _ in range(3): iteration()
opt.enable_timing()
torch.cuda.synchronize(); t0 = time.perf_counter()
marks = []
for _ in range(iters):
    e = torch.cuda.Event(enable_timing=True); e.record(step.stream); marks.append(e); iteration()
e = torch.cuda.Event(enable_timing=True); e.record(step.stream); marks.append(e)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'pretend world {N}: GA {ga}; {iters} iterations: {dt / iters * 1e3:.1f} ms per iteration by the wall clock (incl. the closing synchronize)')
print('  per iteration on the data-gradient stream (ms): ' + ' '.join(f'{marks[i].elapsed_time(marks[i + 1]):.1f}' for i in range(iters)))
ts = opt.timing_summary()
print('  ' + ', '.join(f"{k} {v['ms']:.2f} ms" for k, v in ts.items()))
for line in _streams.log: print('  stream log:', line)
dist.destroy_process_group()
