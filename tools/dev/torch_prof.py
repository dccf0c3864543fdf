import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from triton_client_amd.models import MODEL_ZOO
m = MODEL_ZOO["resnet50"](); m.load(0)
x = torch.randn(8, 3, 224, 224, device="cuda", dtype=torch.float16)
with torch.no_grad():
    for _ in range(3): m._forward_eager(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        m._forward_eager(x)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
# which aten op launches the copy kernels
seen = {}
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::_to_copy", "aten::add_", "aten::to"):
        key = (ev.name, str(ev.input_shapes)[:120])
        seen[key] = seen.get(key, 0) + 1
for k, v in sorted(seen.items(), key=lambda kv: -kv[1])[:30]:
    print(v, k)
