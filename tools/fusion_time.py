import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import imf_oracle as O
from imfnet_amd import ops
from imfnet_amd.model import load_model
dev = torch.device("cuda:0")
sd = O.seeded_state_dict(0, with_unused_image_layers=True)
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
model.load_state_dict(sd); model = model.eval().to(dev)
fw = model._fusion_weights()
for n in (413, 1076, 2176, 4400):
    x = torch.randn(n, 256, device=dev); kv = torch.randn(300, 256, device=dev)
    kt = torch.zeros(128, 320, device=dev); kt[:, :300] = kv[:, :128].t()
    vp = torch.zeros(320, 128, device=dev); vp[:300] = kv[:, 128:]
    ktp, vpp = ops.pack_weights(kt), ops.pack_weights(vp)
    ts, tt = [], []
    with torch.no_grad():
        for r in range(8):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(); o = ops.fusion_attention(x, ktp, vpp, 300, 320, fw); e1.record()
            ref = model._fusion_fast(x, kv); e2.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3); tt.append(e1.elapsed_time(e2) * 1e3)
    print(f"n={n}: fused kernel {np.median(ts[2:]):7.1f} us   torch ops {np.median(tt[2:]):7.1f} us   max|d| {float((o - ref).abs().max()):.1e}")
