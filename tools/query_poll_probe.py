"""Does a host thread that polls hipEventQuery on an event that has not completed slow the GPU down?  (torch's ProcessGroupNCCL
watchdog polls the end events of outstanding collectives.)  Two-stream micro-steps, with and without a polling thread whose event sits
BEHIND the micro-steps on a third stream.  POLL_US = sleep between polls (default 1000)."""
import os, sys, time, threading, torch
sys.path.insert(0, '.')
import bench
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
dev = torch.device('cuda', 0)
unet = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(unet)
batch = bench.synthetic_batch(0, 0, 0, 4, dev)
step = TrainStep(unet, mode='epsilon', grad_accum=8, use_graph=False)
for _ in range(3): step.micro_step(*batch)
step.synchronize()
third = torch.cuda.Stream(dev)
def run(poll_us):
    stop = threading.Event()
    evs = []
    def poller():
        n = 0
        while not stop.is_set():
            for e in list(evs):
                e.query(); n += 1
            time.sleep(poll_us * 1e-6)
        print(f'   ({n} queries)', flush=True)
    th = None
    if poll_us is not None:
        th = threading.Thread(target=poller); th.start()
    ts = []
    for m in range(6):
        a = torch.cuda.Event(enable_timing=True); a.record(step.stream)
        step.micro_step(*batch)
        b = torch.cuda.Event(enable_timing=True); b.record(step.stream)
        # an event on a third stream that completes only when this micro-step has: stays outstanding while the GPU works
        third.wait_stream(step.stream)
        e = torch.cuda.Event(); e.record(third); evs.append(e)
        ts.append((a, b))
    torch.cuda.synchronize(); stop.set()
    if th: th.join()
    print(f'poll every {poll_us} us: ' + ' '.join(f'{a.elapsed_time(b):.1f}' for a, b in ts), flush=True)
run(None); run(100000); run(1000); run(100); run(None)
