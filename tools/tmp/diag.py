import sys, importlib, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_fastpath_model as T
rg = T.rg; L = T.L; lib = L.lib()
m, sd, y, c, tgt = T.build("ti_d2_b64", torch.bfloat16)
m.train()
def grads():
    m.zero_grad()
    logits = m(y, c)
    st = logits.grad_fn.st
    rg.cls_transforms.cross_entropy(logits, tgt, grad_dtype=torch.bfloat16).backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters()}
f = grads()
L.check(lib.rgbnm_set_option(b"mlp_bwd", 0))
p = grads()
for n in f:
    d = (f[n].float() - p[n].float()).abs()
    nd = int((d > 0).sum())
    if nd:
        print(f"{n:50s} differing {nd:8d}/{d.numel():8d}  max|d| {float(d.max()):.3e}  max|v| {float(p[n].abs().max()):.3e}")
p2 = grads()
print("unfused twice identical:", all(torch.equal(p[n], p2[n]) for n in p))
L.check(lib.rgbnm_set_option(b"mlp_bwd", 1))
f2 = grads()
print("fused twice identical:", all(torch.equal(f[n], f2[n]) for n in f))
