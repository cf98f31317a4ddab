import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
import torch_nf_amd as tnf
D, S, L, U, N = 64, 4, 2, 15, 1 << 19
torch.manual_seed(0); np.random.seed(0)
nf = tnf.NormFlow(D, False, "coupling", S, L, U)
with torch.no_grad(): nf(4096)
z = torch.randn(1, N, D, device="cuda")
opt = torch.optim.Adam([nf.params], lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    loss = -nf.log_prob(z).mean()
    loss.backward()
    opt.step()
    return loss
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("train step N=2^19: %.2f ms -> %.1f M samples/s, loss %.4f, peak mem %.2f GB" % (dt * 1e3, N / dt / 1e6, l.item(), torch.cuda.max_memory_allocated() / 2**30))
