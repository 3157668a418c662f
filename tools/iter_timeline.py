"""GPU timeline of one bench iteration (N=1): duration of each of the 8 micro-steps and of the optimizer boundary, from
events on the data-gradient stream.  OVERLAP=0: the update of all regions and the W^T refresh on the main stream (round 3).
PG_INIT=eager|lazy|warm: create the nccl process group although the schedule is local; LATE_LINK_STREAMS=1: the m / v copy streams are
first used AFTER that (what streams.host_link_streams exists to avoid).
EXCHANGE=1: the N > 1 schedule (reduce-scatter from the backward's hooks, sharded update, all-gather under the next forward) over
RCCL in a group of one rank."""
import sys, time, torch
sys.path.insert(0, '.')
import os as _os
from aozora_sdxl_training_amd import _lib as _L
if _os.environ.get('AZ_LIB'):
    _L.LIB_PATH = _os.path.abspath(_os.environ['AZ_LIB'])      # A/B of two builds of the library (tools/ab_iter.sh)
import bench
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
from aozora_sdxl_training_amd.dist import ShardedRaven
dev = torch.device('cuda', 0)
unet = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(unet)
batch = bench.synthetic_batch(0, 0, 0, 4, dev)
ga = 8
step = TrainStep(unet, mode='epsilon', grad_accum=ga, use_graph=False)
import os
EXCH = os.environ.get('EXCHANGE', '0') == '1'       # the data-parallel schedule over RCCL in a group of ONE rank (collectives = identity)
from aozora_sdxl_training_amd import streams as _streams
if os.environ.get('LATE_LINK_STREAMS') != '1':      # as bench.py / trainer.main do: the m / v copy streams make their first copies before the communicator exists
    _streams.host_link_streams(dev)
PG = os.environ.get('PG_INIT', '')          # 'eager' / 'lazy': create the process group although the schedule is local
if EXCH or PG:
    import socket, torch.distributed as dist
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
    if PG == 'lazy': dist.init_process_group(backend='nccl')
    else: dist.init_process_group(backend='nccl', device_id=dev)
    if PG == 'warm':
        t_ = torch.zeros(8, device=dev); dist.all_reduce(t_); torch.cuda.synchronize()
opt = ShardedRaven(unet, lr=8e-7, clip_grad_norm=1.0, overlap=os.environ.get('OVERLAP', '1') == '1', force_exchange=EXCH)
for _ in range(3): step.micro_step(*batch)
step.synchronize(); opt.zero_grad()
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(step.stream); return e
host = []
def iteration(marks):
    for m in range(ga):
        if m == ga - 2: opt.prefetch()
        marks.append(ev()); step.micro_step(*batch, after_tail=opt.reduce_tail if (EXCH and m == ga - 1 and opt.overlap and os.environ.get('NO_HOOK') != '1') else None); host.append(time.perf_counter())
    marks.append(ev())
    with torch.cuda.stream(step.stream):
        pass
    opt.step(); host.append(time.perf_counter()); marks.append(torch.cuda.Event(enable_timing=True)); marks[-1].record(torch.cuda.current_stream())
    opt.zero_grad()
iteration([]); torch.cuda.synchronize()
opt.enable_timing()
for it in range(int(os.environ.get('ROUNDS', '2'))):
    marks = []; del host[:]; t0 = time.time(); h0 = time.perf_counter(); iteration(marks); marks2 = []; iteration(marks2); torch.cuda.synchronize(); t1 = time.time()
    d = [marks[i].elapsed_time(marks[i + 1]) for i in range(ga)]
    print('host: micro-step calls returned at (ms): ' + ' '.join(f'{(h - h0) * 1e3:.0f}' for h in host[:ga + 1]) + '  (9th = optimizer step issued)', flush=True)
    print('micro-steps (ms): ' + ' '.join(f'{x:.1f}' for x in d), flush=True)
    print(f'  last micro-step end -> optimizer kernels done on the default stream: {marks[ga].elapsed_time(marks[ga + 1]):.1f} ms;'
          f' -> first micro-step of the next iteration starts: {marks[ga].elapsed_time(marks2[0]):.1f} ms; 2 iterations wall {1e3 * (t1 - t0):.0f} ms', flush=True)
torch.cuda.synchronize()
print('spans of the exchange / copies, ms since the start of the last pair of iterations (first micro-step = 0):')
for name, nbytes, e0, e1 in (opt._timing or []):
    a, b = marks[0].elapsed_time(e0), marks[0].elapsed_time(e1)
    if a >= -1.0:
        print(f'   {name:36s} {a:8.1f} -> {b:8.1f}', flush=True)
ts = opt.timing_summary()
for line in _streams.log: print('stream log:', line)
print({k: (round(v['ms'], 1), v['calls']) for k, v in ts.items()}, flush=True)
if os.environ.get('PROBE_PAIRS', '0') == '1':     # which of the streams share a hardware queue (side-by-side spin: ~1.1 apart, ~2 on one queue)
    from aozora_sdxl_training_amd import streams as S
    named = [('default', torch.cuda.default_stream(dev)), ('data-gradient', step.stream), ('weight-gradient', unet._sides[0]),
             ('exchange', opt.comm), ('h2d', opt.copy_streams[0]), ('d2h', opt.copy_streams[1])]
    for i in range(len(named)):
        for j in range(i + 1, len(named)):
            if named[i][1].cuda_stream != named[j][1].cuda_stream:
                print(f'  {named[i][0]:16s} / {named[j][0]:16s}: side-by-side {S.spin_pair_ratio(named[i][1], named[j][1]):.2f}x', flush=True)
