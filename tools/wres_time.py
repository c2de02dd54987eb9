import sys, time, torch
sys.path.insert(0, '.')
from rgb_no_more_amd import lib as L
M, N, epi = 50176, 768, 2
A = torch.randn(M, 192, device="cuda").bfloat16(); W = torch.randn(N, 192, device="cuda").bfloat16() * 0.1
b = torch.randn(N, device="cuda"); Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); C2 = torch.empty_like(Cc)
def run():
    L.check(L.lib().rgbnm_gemm_nt(1, epi, A.data_ptr(), 192, W.data_ptr(), 192, Cc.data_ptr(), N, b.data_ptr(), None, 0,
                                  C2.data_ptr(), N, None, 0, M, N, 192, 0, L.stream()))
t_end = time.time() + 2.0
while time.time() < t_end:
    for _ in range(50): run()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(300): run()
e1.record(); torch.cuda.synchronize()
print(sys.argv[1], 'GELU kernel us:', e0.elapsed_time(e1) * 1000 / 300)
