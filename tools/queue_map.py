"""Which of the package's streams share a hardware queue: pairs of long one-lane kernels on two streams take T when the streams sit on
different queues and 2T when they share one (HIP multiplexes streams onto four queues).   usage: python tools/queue_map.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from imfnet_amd import ops, _lib
dev = torch.device("cuda:0")
raw, views = ops.aux_streams(dev)
L = _lib.lib()
extra = [torch.cuda.ExternalStream(L.imf_stream_create(), device=dev) for _ in range(4)]
names = ["main", "side", "image"] + ["extra%d" % i for i in range(4)]
streams = list(views) + extra
def t_pair(a, b):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(a): torch.cuda._sleep(20_000_000)
    with torch.cuda.stream(b): torch.cuda._sleep(20_000_000)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
t_pair(streams[0], streams[1])
base = t_pair(streams[0], streams[0])
print("same stream twice: %.2f ms" % base)
for i in range(len(streams)):
    print(names[i].ljust(8), " ".join("%6.2f" % t_pair(streams[i], streams[j]) if i != j else "   -  " for j in range(len(streams))))
