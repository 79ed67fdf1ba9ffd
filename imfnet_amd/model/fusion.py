"""Point <-> image cross-attention fusion block (the 'AF' of IMFNet).

Interface parity with the reference's model/attention_fusion.py: class names, constructor
arguments, `forward(data, mask=None, queries_encoder=None)` and the parameter names
(`cross_attend_blocks.0.fn.to_q.weight`, `...0.norm_context.weight`, `...1.fn.net.2.bias`, ...)
are identical, so a reference checkpoint loads with strict=True.

Arithmetic (attention_fusion.py:65-95,132-154): pre-LayerNorm single-head cross attention of the
N stride-8 point features (dim 256) over the 300 image tokens (dim 128), softmax over the tokens,
residual; then a pre-LayerNorm GEGLU feed-forward (256 -> 2x1024 -> 256), residual.  All fp32.
These are dense contractions and run through PyTorch-ROCm (hipBLASLt); they are ~5 % of the
path's FLOPs.
"""
import torch
import torch.nn.functional as F
from torch import nn


class PreNorm(nn.Module):
    def __init__(self, dim, fn, context_dim=None):
        super().__init__()
        self.fn = fn
        self.norm = nn.LayerNorm(dim)
        self.norm_context = nn.LayerNorm(context_dim) if context_dim is not None else None

    def forward(self, x, **kwargs):
        x = self.norm(x)
        if self.norm_context is not None:
            kwargs['context'] = self.norm_context(kwargs['context'])
        return self.fn(x, **kwargs)


class GEGLU(nn.Module):
    def forward(self, x):
        value, gate = x.chunk(2, dim=-1)
        return value * F.gelu(gate)            # exact (erf) GELU


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, dim * mult * 2), GEGLU(), nn.Linear(dim * mult, dim))

    def forward(self, x):
        return self.net(x)


class Attention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_kv = nn.Linear(query_dim if context_dim is None else context_dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, query_dim)

    def forward(self, x, context=None, mask=None):
        ctx = x if context is None else context
        b, n, h, d = x.shape[0], x.shape[1], self.heads, self.dim_head
        q = self.to_q(x).view(b, n, h, d).transpose(1, 2)                   # [b,h,n,d]
        k, v = self.to_kv(ctx).chunk(2, dim=-1)                              # K first, V second
        k = k.reshape(b, -1, h, d).transpose(1, 2)
        v = v.reshape(b, -1, h, d).transpose(1, 2)
        sim = torch.matmul(q, k.transpose(-1, -2)) * self.scale
        if mask is not None:
            sim = sim.masked_fill(~mask.reshape(b, 1, 1, -1), -torch.finfo(sim.dtype).max)
        out = torch.matmul(sim.softmax(dim=-1), v)                           # [b,h,n,d]
        return self.to_out(out.transpose(1, 2).reshape(b, n, h * d))


class AttentionFusion(nn.Module):
    def __init__(self, depth, dim, latent_dim=512, cross_heads=1, latent_heads=8, cross_dim_head=64,
                 latent_dim_head=64, weight_tie_layers=False):
        super().__init__()
        self.cross_attend_blocks = nn.ModuleList([
            PreNorm(latent_dim, Attention(latent_dim, dim, heads=cross_heads, dim_head=cross_dim_head),
                    context_dim=dim),
            PreNorm(latent_dim, FeedForward(latent_dim)),
        ])
        self.layers = nn.ModuleList([])
        tied = None
        for _ in range(depth):                     # depth == 0 on IMFNet's path (resunet.py:93)
            if tied is None or not weight_tie_layers:
                tied = nn.ModuleList([
                    PreNorm(latent_dim, Attention(latent_dim, heads=latent_heads, dim_head=latent_dim_head)),
                    PreNorm(latent_dim, FeedForward(latent_dim))])
            self.layers.append(tied)

    def forward(self, data, mask=None, queries_encoder=None):
        x = queries_encoder
        attn, ff = self.cross_attend_blocks
        x = attn(x, context=data, mask=mask) + x
        x = ff(x) + x
        for self_attn, self_ff in self.layers:
            x = self_attn(x) + x
            x = self_ff(x) + x
        return x
