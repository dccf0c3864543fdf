import sys, torch
sys.path.insert(0, '.')
from triton_client_amd.ops.copy import batched_copy
n = 1204224
src = [torch.randint(0, 255, (n,), device='cuda', dtype=torch.uint8) for _ in range(16)]
dst = torch.zeros(16 * n, device='cuda', dtype=torch.uint8)
tri = [(s.data_ptr(), dst.data_ptr() + i * n, n) for i, s in enumerate(src)]
small = [(dst.data_ptr() + i * 32000, src[i].data_ptr(), 32000) for i in range(16)]
for name, t in (('gather 16 x 1.2 MB', tri), ('scatter 16 x 32 KB', small)):
    for _ in range(5): batched_copy(t)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): batched_copy(t)
    b.record(); torch.cuda.synchronize()
    print(f'{name}: {a.elapsed_time(b) * 1e3 / 200:.1f} us per launch')
