import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_fastpath_model as T
rg = T.rg; L = T.L; lib = L.lib()
T.CASES["ti_d1_b64"] = (192, 3, 1, 64, False)
m, sd, y, c, tgt = T.build("ti_d1_b64", torch.bfloat16)
m.train()
def run():
    m.zero_grad()
    logits = m(y, c)
    ar = logits.grad_fn.st.arena
    rg.cls_transforms.cross_entropy(logits, tgt, grad_dtype=torch.bfloat16).backward()
    torch.cuda.synchronize()
    return {"du": ar.du.clone(), "dx_mid": ar.dx_mid.clone(), "x_mid": ar.blk[0]["x_mid"].clone(), "mean2": ar.blk[0]["mean2"].clone(),
            "rstd2": ar.blk[0]["rstd2"].clone(), "gp": ar.blk[0]["u"].clone()}
f = run()
L.check(lib.rgbnm_set_option(b"mlp_bwd", 0))
p = run()
for k in f:
    d = (f[k].float() - p[k].float()).abs()
    nz = (d > 0).nonzero()
    print(k, "differing", len(nz), "of", d.numel(), "max", float(d.max()))
    if len(nz) and k == "dx_mid":
        rows = nz[:, 0]
        print(" rows mod 49:", torch.unique(rows % 49).tolist())
        print(" cols:", torch.unique(nz[:, 1]).tolist()[:40])
        print(" first:", nz[:10].tolist())
        for q in range(min(5, len(nz))):
            i, j = nz[q].tolist()
            print(" values", float(f[k][i, j]), float(p[k][i, j]))
L.check(lib.rgbnm_set_option(b"ln_fuse", 0))
q = run()
print("dxn available:", hasattr(logits_ar := None, "x"))
for name, a, b in (("fused-mlp vs separate", f, q), ("kpipe-lnbwd vs separate", p, q)):
    d = (a["dx_mid"].float() - b["dx_mid"].float()).abs()
    print(name, "dx_mid differing", int((d > 0).sum()))
