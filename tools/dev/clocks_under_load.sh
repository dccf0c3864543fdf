#!/bin/bash
# Shader clock, memory clock and power of GPU 0 while two ResNet-50 instances replay batch-128 graphs back to
# back (tools/dev/instance_overlap.py) and while hipBLASLt multiplies 8192^3 matrices; idle first.
set -u
cd "$(dirname "$0")/../.."
sample() { rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | sed 's/^/    /'; }
echo "idle:"; sample
timeout -k 10 120 python -u tools/dev/instance_overlap.py > /tmp/overlap.log 2>&1 &
PID=$!
echo "two ResNet-50 instances (model load first; one sample every 3 s until the run ends):"
while kill -0 $PID 2>/dev/null; do rocm-smi -d 0 --showclocks 2>/dev/null | grep sclk | sed 's/^/    /'; sleep 3; done
wait $PID
tail -n 3 /tmp/overlap.log
timeout -k 10 60 python -u - <<'PY' &
import time, torch
a = torch.randn(8192, 8192, device="cuda").half(); b = torch.randn(8192, 8192, device="cuda").half()
t0 = time.time()
while time.time() - t0 < 12:
    for _ in range(50):
        a @ b
    torch.cuda.synchronize()
PY
PID=$!
sleep 8
echo "hipBLASLt 8192^3 loop:"; sample
wait $PID
