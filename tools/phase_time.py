"""Un-profiled duration of the phases of a training step (forward / loss / backward / Adam) from HIP events at the phase
boundaries only, against the kernel-time sums a rocprofv3 trace gives for the same phases: what the kernel boundaries cost
without the profiler.  usage: python tools/phase_time.py [--dtype bf16] [--steps 40] [--fwd-only]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from floodplanet_code_amd.unet import HipUNet

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--batch", type=int, default=16); ap.add_argument("--size", type=int, default=256)
args = ap.parse_args()
dev = torch.device("cuda:0")
net = HipUNet(8, 2, bilinear=True, precision={"bf16": "bf16", "f16": "fp16", "f32": "fp32"}[args.dtype]).to(dev)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.rand(args.batch, 8, args.size, args.size, device=dev, generator=g)
t = (torch.rand(args.batch, args.size, args.size, device=dev, generator=g) > 0.5).long()
net.train()
def step(ev=None):
    if ev: ev[0].record()
    net._forward_raw(x, True, want_logits=False)
    if ev: ev[1].record()
    net._loss_raw(t, 0, dev)
    if ev: ev[2].record()
    net._backward_raw(None, dev)
    if ev: ev[3].record()
    step.n += 1
    net.adam_step(1e-4, step.n, (0.9, 0.999), 1e-8)
    if ev: ev[4].record()
step.n = 0
for _ in range(5): step()
torch.cuda.synchronize()
acc = [0.0] * 4
evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
for i in range(args.steps): step(evs[i])
torch.cuda.synchronize()
for e in evs:
    for k in range(4): acc[k] += e[k].elapsed_time(e[k + 1])
tot = evs[0][0].elapsed_time(evs[-1][4]) / args.steps
print("phase ms: forward %.3f  loss %.3f  backward %.3f  adam(+pack at next forward) %.3f  | step %.3f" %
      tuple([a / args.steps for a in acc] + [tot]))
# forward only, back to back (no loss / backward): the kernel-boundary cost of the 18 conv + 18 statistics launches
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(args.steps): net._forward_raw(x, True, want_logits=False)
b.record(); torch.cuda.synchronize()
print("forward only, back to back: %.3f ms" % (a.elapsed_time(b) / args.steps))
