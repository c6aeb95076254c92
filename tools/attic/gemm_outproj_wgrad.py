import torch
torch.manual_seed(0)
B,C,L,D=8,1024,8192,1024
y=torch.randn(B,C,L,device="cuda",dtype=torch.bfloat16)
dout=torch.randn(B,L,D,device="cuda",dtype=torch.bfloat16)
def t(f,n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1000
a=lambda: torch.bmm(y,dout).sum(0,dtype=torch.float32).t().contiguous()
b=lambda: torch.bmm(dout.transpose(1,2),y.transpose(1,2)).sum(0,dtype=torch.float32)
a1=lambda: torch.bmm(y,dout)
b1=lambda: torch.bmm(dout.transpose(1,2),y.transpose(1,2))
# warm clocks
for _ in range(50): a1()
print("A: bmm(y,dout).sum.t.contig   %.1f us (bmm alone %.1f)"%(t(a),t(a1)))
print("B: bmm(dout^T,y^T).sum        %.1f us (bmm alone %.1f)"%(t(b),t(b1)))
print("A again %.1f  B again %.1f"%(t(a),t(b)))
print((a()-b()).abs().max().item(), a().abs().max().item())
