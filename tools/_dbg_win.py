import sys, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import rgb_no_more_amd as rg
from rgb_no_more_amd import swinv2 as SW, detfill
DEV="cuda"
B,res,heads,shift=1,8,1,0
C_=heads*32
for dt in (torch.float32, torch.bfloat16):
    qkv = torch.from_numpy(detfill.normalish((B*res*res, 3*C_), 21)).to(DEV).to(dt)
    bias = torch.zeros(heads,64,64, device=DEV)
    scale = torch.full((heads,), 10.0, device=DEV)
    out = SW._WinAttnFn.apply(qkv, bias, scale, B, res, C_, heads, shift)
    x = qkv.float().cpu().reshape(64, 3, heads, 32)
    q,k,v = x[:,0,0], x[:,1,0], x[:,2,0]
    att = torch.nn.functional.normalize(q,dim=-1) @ torch.nn.functional.normalize(k,dim=-1).T * 10.0
    ref = torch.softmax(att,-1) @ v
    d = (out.float().cpu() - ref).abs()
    print(dt, "max err", d.max().item())
    print(" per-token err (first 16):", d.max(1).values[:16].numpy().round(3))
    print(" per-dim err:", d.max(0).values.numpy().round(3))
    # test: V only (uniform attention) -> scale 0
    out0 = SW._WinAttnFn.apply(qkv, bias, scale*0, B, res, C_, heads, shift)
    print(" uniform-attn err per dim:", (out0.float().cpu() - v.mean(0, keepdim=True)).abs().max(0).values.numpy().round(3))
