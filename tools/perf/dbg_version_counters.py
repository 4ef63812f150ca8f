import torch
dev = "cuda"
arena = torch.nn.Parameter(torch.randn(1000, device=dev))
base = arena.detach()
view = torch.nn.Parameter(base[10:110].view(10, 10))
arena.grad = torch.randn_like(arena)
for fused in (True, False):
    opt = torch.optim.Adam([arena], lr=1e-3, fused=fused)
    v0 = (arena._version, base._version, view._version)
    before = view.detach().clone()
    opt.step()
    print("fused", fused, "versions before", v0, "after", (arena._version, base._version, view._version), "values changed", not torch.equal(before, view.detach()))
sep = [torch.nn.Parameter(torch.randn(100, device=dev)) for _ in range(3)]
for p in sep:
    p.grad = torch.randn_like(p)
opt = torch.optim.Adam(sep, lr=1e-3, fused=True)
v0 = [p._version for p in sep]
opt.step()
print("separate fused", v0, [p._version for p in sep])
