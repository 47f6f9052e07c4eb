import torch, time
w = torch.empty(1024*256*3, device='cuda')
def t(fn, n=2000):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter()-t0)/n*1e6
print('slice+view', t(lambda: w[:1024*256].view(256,1024)))
print('slice only', t(lambda: w[:1024*256]))
v = w[:1024*256]
print('view only', t(lambda: v.view(256,1024)))
print('empty', t(lambda: torch.empty(18063, 1024, device='cuda')))
print('empty small', t(lambda: torch.empty(2, 1, 256, device='cuda')))
x = torch.empty(2,1,256, device='cuda')
print('index x[0]', t(lambda: x[0]))
print('data_ptr', t(lambda: w.data_ptr()))
# inside autograd function?
class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        return a[:1024*256].view(256,1024).clone()
    @staticmethod
    def backward(ctx, g): return None
a = w.clone().requires_grad_(True)
print('in Function.apply (incl clone kernel)', t(lambda: F.apply(a), 500))
class G(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, wts):
        b = wts[:1024*256].view(256,1024)
        c = wts[1024*256:2*1024*256].view(1024,256)
        ctx.save_for_backward(b, c)
        return a * 1
    @staticmethod
    def backward(ctx, g): return g, None
print('Function with 2 views of a non-grad buffer saved', t(lambda: G.apply(a, w), 500))
