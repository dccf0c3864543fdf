"""GPU vision models: ResNet-50 (random-init, fp16, channels-last) for MI355X.

The network is defined here (torchvision is not part of the image) with the
standard bottleneck layout (3-4-6-3, stride on the 3x3 conv).  For serving:

* BatchNorm is folded into the preceding convolution at load time, so the
  executed graph is conv(+bias)+ReLU only;
* weights are fp16 / channels-last (NHWC is what MIOpen's MFMA kernels want);
* the whole forward is captured in a HIP graph per (batch size, input address,
  output address) – a request whose tensors live in hipshm regions replays a
  graph that reads the input region and writes the output region directly, so
  there is no staging copy at all (the reference's CUDA-shm path always stages
  through the framework's own buffers).

Model contract (what ``image_client`` expects, reference
``src/python/examples/image_client.py:59-150``): one ``FP16``/``FP32`` NCHW
input ``[3,224,224]`` with batch dimension, one ``FP32`` vector output
``[1000]``.
"""

from __future__ import annotations

import os
import threading
from collections import OrderedDict

from ..utils import trace as _trace

from .base import Model, ModelError, TensorSpec, register

def install_miopen_db():
    """Point MIOpen at the tuned find-db / perf-db shipped next to this file.

    MIOpen's exhaustive search costs 7-13 s per new convolution shape set on
    MI355X and a fresh machine has no user database, so an untuned server runs
    the batch-8 ResNet-50 forward in 1.65 ms instead of 1.03 ms.  The databases
    in ``miopen_db/`` were produced by ``tools/tune_miopen.py`` on gfx950 for
    every batch size the batcher can form; they are copied to a private
    writable directory (MIOpen appends to its user db) and selected through
    ``MIOPEN_USER_DB_PATH`` unless the user already chose one.  Must run before
    the first convolution of the process.
    """
    import os
    import shutil
    import tempfile

    if os.environ.get("MIOPEN_USER_DB_PATH"):
        return os.environ["MIOPEN_USER_DB_PATH"]
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
    dst = os.path.join(tempfile.gettempdir(), f"tca_miopen_db_{os.getuid()}_{os.getpid()}")
    os.makedirs(dst, exist_ok=True)
    if os.path.isdir(src):
        for name in os.listdir(src):
            if name.endswith(".txt"):
                shutil.copy(os.path.join(src, name), os.path.join(dst, name))
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst


_LAYERS = (3, 4, 6, 3)
_WIDTHS = (64, 128, 256, 512)


def _build_resnet50(torch, num_classes=1000):
    nn = torch.nn

    class ConvBN(nn.Module):
        def __init__(self, cin, cout, k, stride=1, relu=True):
            super().__init__()
            self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)
            self.bn = nn.BatchNorm2d(cout)
            self.relu = relu

        def forward(self, x):
            x = self.bn(self.conv(x)) if self.bn is not None else self.conv(x)
            return torch.relu_(x) if self.relu else x

        def fold(self):
            """Fold the BatchNorm affine transform into the convolution."""
            bn = self.bn
            if bn is None:
                return
            w = self.conv.weight
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            fused = nn.Conv2d(self.conv.in_channels, self.conv.out_channels, self.conv.kernel_size,
                              stride=self.conv.stride, padding=self.conv.padding, bias=True)
            fused.weight.data = (w * scale.reshape(-1, 1, 1, 1)).to(w.dtype)
            fused.bias.data = (bn.bias - bn.running_mean * scale).to(w.dtype)
            self.conv = fused
            self.bn = None

    class Bottleneck(nn.Module):
        def __init__(self, cin, width, stride):
            super().__init__()
            cout = width * 4
            self.a = ConvBN(cin, width, 1)
            self.b = ConvBN(width, width, 3, stride=stride)
            self.c = ConvBN(width, cout, 1, relu=False)
            self.down = ConvBN(cin, cout, 1, stride=stride, relu=False) if (stride != 1 or cin != cout) else None

        def forward(self, x):
            identity = x if self.down is None else self.down(x)
            y = self.c(self.b(self.a(x)))
            y += identity
            return torch.relu_(y)

    class ResNet50(nn.Module):
        def __init__(self):
            super().__init__()
            self.stem = ConvBN(3, 64, 7, stride=2)
            self.pool = nn.MaxPool2d(3, stride=2, padding=1)
            blocks = []
            cin = 64
            for stage, (n, width) in enumerate(zip(_LAYERS, _WIDTHS)):
                for i in range(n):
                    stride = 2 if (i == 0 and stage > 0) else 1
                    blocks.append(Bottleneck(cin, width, stride))
                    cin = width * 4
            self.blocks = nn.Sequential(*blocks)
            self.fc = nn.Linear(cin, num_classes)

        def features(self, x):
            x = self.pool(self.stem(x))
            x = self.blocks(x)
            return x.mean(dim=(2, 3))  # global average pool -> [B, 2048]

        def forward(self, x):
            return self.fc(self.features(x))

        def fold_batchnorm(self):
            for m in self.modules():
                if isinstance(m, ConvBN):
                    m.fold()

    return ResNet50()


def _is_channels_last(torch, t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


class _GraphCache:
    """LRU of captured HIP graphs keyed by (batch, in_ptr, out_ptr, instance).

    A graph that leaves the cache (LRU eviction, or its buffers' region being unmapped) may
    still be enqueued on its instance's stream: it is parked together with an event recorded
    on that stream and destroyed only once the event has completed."""

    def __init__(self, capacity=512):
        self._items = OrderedDict()
        self._capacity = capacity
        self._parked = []  # [(event, graph entry)]

    def __len__(self):
        return len(self._items)

    def get(self, key):
        item = self._items.get(key)
        if item is not None:
            self._items.move_to_end(key)
        return item

    def put(self, key, value, retire=None):
        """``retire(key) -> event`` records an event behind everything enqueued for ``key``."""
        self._items[key] = value
        while len(self._items) > self._capacity:
            old_key, old = self._items.popitem(last=False)
            self._park(old_key, old, retire)
        self.collect()

    def evict_if(self, predicate, retire=None):
        for key in [k for k, v in self._items.items() if predicate(k, v)]:
            self._park(key, self._items.pop(key), retire)
        self.collect()

    def _park(self, key, value, retire):
        event = retire(key) if retire is not None else None
        if event is None:
            return  # nothing can be in flight: drop it now
        self._parked.append((event, value))

    def collect(self):
        self._parked = [(e, v) for e, v in self._parked if not e.query()]


# GPU models of this process by name: ensembles resolve their steps here
_LOADED = {}


@register
class ResNet50(Model):
    name = "resnet50"
    platform = "pytorch_rocm"
    backend = "pytorch"
    device = "gpu"
    max_batch_size = 128
    dynamic_batching = True
    max_queue_delay_us = 0  # launch as soon as the GPU has room; batches form while it is busy
    # ... and while every execution instance still has a batch to run, a batch that is not full may wait this
    # long for more requests (measured at concurrency 64: every merged batch full instead of four in five,
    # 74-75k -> 77.5k infer/s, p99 10.0 -> 7.0 ms)
    fill_window_us = 600
    # batch sizes whose HIP graphs are captured at load time (the batcher merges
    # concurrent requests, so totals other than the request batch show up)
    warmup_batches = (1, 2, 4, 8, 16, 24, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120, 128)
    pad_batches = True  # other totals run on the next of these sizes: nothing is captured while serving
    # Served by the native executor from captured graphs (requests with both tensors in device shared
    # memory never reach the interpreter): the batch sizes it holds a graph for.  A merged batch runs on
    # the smallest graph that holds it, so every request batch size - 1 included, perf_analyzer's
    # default - is served from graphs captured at load time and nothing is ever captured online.
    native_kernel = "graph"
    native_batches = (1, 2, 4, 8, 16, 24, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120, 128)
    inputs = [TensorSpec("input", "FP16", [3, 224, 224], format="FORMAT_NCHW")]
    outputs = [TensorSpec("output", "FP32", [1000], label_filename="labels.txt")]

    def __init__(self):
        super().__init__()
        if os.environ.get("TCA_RESNET_MAX_BATCH"):  # experiments with the batcher's largest merged batch
            self.max_batch_size = int(os.environ["TCA_RESNET_MAX_BATCH"])
            self.warmup_batches = (1, 2, 4) + tuple(range(8, self.max_batch_size + 1, 8))
        self._torch = None
        self._net = None
        self._graphs = _GraphCache(int(os.environ.get("TCA_RESNET_GRAPH_CACHE", "512")))
        self._io_buffers = {}  # (instance, batch) -> static (input, output) pair for callers without stable buffers
        # entered around every stream capture: the server installs the native executor's gate here, which keeps
        # that executor's own graph launches (other threads, same streams) out of the way of a capture
        self.capture_gate = None
        self.online_captures = 0  # graphs captured after the server finished loading the model (``warmed_up``)
        self.warmed_up = False
        self._lock = threading.RLock()
        self._stream = None
        self.labels = [f"class_{i}" for i in range(1000)]
        self.use_graphs = True
        # hand-written fused kernels (default) vs the plain library forward
        self.fused = os.environ.get("TCA_RESNET_FUSED", "1") != "0"
        # conv 7x7 + bias + ReLU + max pool as one kernel (A/B switch; off = library conv + fused pool pass)
        self.fused_stem = os.environ.get("TCA_RESNET_FUSED_STEM", "1") != "0"
        # projection shortcuts on a parallel graph branch.  Off: measured on MI355X the
        # cross-queue dependencies of a forked HIP graph cost more than the overlap wins
        # (b8 0.590 vs 0.526 ms, b128 2.626 vs 2.614 ms); kept as an A/B switch.
        self.overlap_shortcut = os.environ.get("TCA_RESNET_OVERLAP", "0") != "0"
        # average pooling + fully connected layer on the in-tree kernels (A/B switch)
        self.fused_tail = os.environ.get("TCA_RESNET_FUSED_TAIL", "1") != "0"
        # 3x3 convolution + 1x1 expansion of a bottleneck block as one kernel (A/B switch)
        self.fused_expansion = os.environ.get("TCA_RESNET_FUSED_EXPANSION", "1") != "0"
        # Execution instances: each has its own HIP stream and its own captured graphs
        # (with private activation pools), so the dynamic batcher keeps this many batches
        # running side by side.  The tail of one batch's layers (fewer workgroups than the
        # 256 CUs in the late stages) is filled by the other batch's kernels: two instances
        # measured +13% inferences/sec over one at batch 128 on one MI355X.
        self.instance_count = max(1, int(os.environ.get("TCA_RESNET_INSTANCES", "2")))

    # -- lifecycle --
    def load(self, device_index=None):
        import os

        install_miopen_db()
        import torch

        if not torch.cuda.is_available():
            raise ModelError("resnet50 needs an MI355X (torch.cuda.is_available() is False)")
        self._torch = torch
        self._device = torch.device("cuda", device_index or 0)
        torch.cuda.set_device(self._device)
        gen = torch.Generator().manual_seed(0)
        net = _build_resnet50(torch)
        # random-init weights of the real architecture (no checkpoints offline):
        # keep activations in a sane range so fp16 does not overflow
        with torch.no_grad():
            for p in net.parameters():
                if p.dim() > 1:
                    fan_in = p[0].numel()
                    p.copy_(torch.randn(p.shape, generator=gen) * (1.0 / fan_in) ** 0.5)
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_var.fill_(1.0)
                    m.running_mean.zero_()
                elif getattr(m, "bias", None) is not None and m.bias.dim() == 1:
                    # from the same generator: every replica of the model (other processes,
                    # other GPUs) holds the same weights
                    m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.01)
        net.eval()
        net.fold_batchnorm()
        net = net.to(self._device, dtype=torch.float16).to(memory_format=torch.channels_last)
        for p in net.parameters():
            p.requires_grad_(False)
        self._net = net
        self._build_plan()
        self._streams = [torch.cuda.Stream(device=self._device) for _ in range(self.instance_count)]
        self._stream = self._streams[0]
        self._graph_pools = [None] * self.instance_count
        self._side_stream = torch.cuda.Stream(device=self._device)
        _LOADED[self.name] = self
        # immediate mode + the shipped tuned databases; TCA_MIOPEN_TUNE=1 runs
        # MIOpen's exhaustive search instead (tools/tune_miopen.py)
        torch.backends.cudnn.benchmark = os.environ.get("TCA_MIOPEN_TUNE", "0") == "1"
        # warm-up: lets MIOpen pick its kernels before the first request / capture
        x = torch.zeros(8, 3, 224, 224, device=self._device, dtype=torch.float16)
        with torch.no_grad(), torch.cuda.stream(self._stream):
            for _ in range(2):
                self._forward_eager(x)
        self._stream.synchronize()

    def unload(self):
        if _LOADED.get(self.name) is self:
            del _LOADED[self.name]
        self._graphs = _GraphCache(int(os.environ.get("TCA_RESNET_GRAPH_CACHE", "512")))
        self._io_buffers = {}
        self._graph_pools = [None] * len(getattr(self, "_graph_pools", []))
        self._net = None
        # the flattened weights, the packed stem and the per-stream workspaces go with the model
        self._plan = None
        self._proj, self._rows_w2, self._stem_packed = {}, {}, None
        if self._torch is not None:
            from ..ops import conv

            streams = (list(getattr(self, "_streams", [])) + [getattr(self, "_side_stream", None)]
                       + list(getattr(self, "_plan_streams", [])))
            self._plan_streams = []
            for st in streams:
                if st is not None:
                    st.synchronize()
            conv.release_splitk_workspaces(self._device, [st for st in streams if st is not None])
            import gc

            gc.collect()  # captured graphs sit in reference cycles (closures over their static tensors)
            # PyTorch keeps one BLAS workspace (76 MiB on this GPU) per stream that ever ran a library GEMM - the
            # eager warm-up does, on a stream that goes with the model
            clear = getattr(self._torch._C, "_cuda_clearCublasWorkspaces", None)
            if clear is not None:
                clear()
            self._torch.cuda.empty_cache()

    # -- execution --
    def _forward_eager(self, x):
        torch = self._torch
        if self.fused:
            return self._net.fc(self._features_fused(x).mean(dim=(2, 3))).float()
        return self._net(x.contiguous(memory_format=torch.channels_last)).float()

    def _build_plan(self):
        """Flatten the folded network into plain tensors for the fused forward:
        1x1 weights as [Cout, Cin] matrices for the MFMA GEMM, spatial kernels in
        channels-last for the library convolution."""
        torch = self._torch
        net = self._net

        def mat(m):
            w = m.conv.weight
            return (w.reshape(w.shape[0], w.shape[1]).contiguous(), m.conv.bias.contiguous(), m.conv.stride[0])

        def spatial(m):
            return (m.conv.weight.contiguous(memory_format=torch.channels_last), m.conv.bias.contiguous(),
                    m.conv.stride[0], m.conv.padding[0])

        blocks = []
        self._proj = {}  # block index -> ([expansion | projection] weights, summed bias) for the in-kernel shortcut
        self._rows_w2 = {}  # block index -> permuted expansion (| projection) weights of the row-organised kernel
        for index, blk in enumerate(net.blocks):
            blocks.append((mat(blk.a), spatial(blk.b), mat(blk.c), mat(blk.down) if blk.down is not None else None))
            a, mid, c, down = blocks[-1]
            if (down is not None and down[2] == 1 and down[0].shape[1] == 64 and mid[0].shape[0] == 64 and mid[2] == 1
                    and c[2] == 1):
                self._proj[index] = (torch.cat([c[0], down[0]], dim=1).contiguous(),
                                     (c[1].float() + down[1].float()).half().contiguous())
            # expansion weights in the K order of the row-organised block kernel (64 -> 64 -> 256, stride 1)
            w4 = mid[0]
            if tuple(w4.shape) == (64, 64, 3, 3) and mid[2] == 1 and mid[3] == 1 and c[0].shape[0] == 256 and c[2] == 1:
                from ..ops import conv as _conv

                self._rows_w2[index] = _conv.permute_mid_k(self._proj[index][0] if index in self._proj else c[0])
        self._plan = (spatial(net.stem), blocks)
        # the stem also as the packed matrix of the fused conv + bias + ReLU + max-pool kernel
        from ..ops import conv

        w = net.stem.conv
        self._stem_packed = None
        if (self.fused_stem and tuple(w.weight.shape) == (64, 3, 7, 7) and w.stride[0] == 2 and w.padding[0] == 3):
            self._stem_packed = (conv.pack_stem_weight(w.weight), w.bias.contiguous())

    def _features_fused(self, x):
        """ResNet-50 trunk with every bias / residual / ReLU fused: 1x1
        convolutions run on the MFMA GEMM kernel with the epilogue inside,
        3x3 convolutions on the same kernel as an implicit GEMM where the
        layer has enough output tiles to fill the chip, otherwise on the library
        kernel followed by ONE in-place epilogue pass."""
        torch = self._torch
        F = torch.nn.functional
        from ..ops import conv

        def pointwise(t, layer, residual, relu):
            w2, bias, stride = layer
            cout, cin = w2.shape
            pixels = t.shape[0] * ((t.shape[2] - 1) // stride + 1) * ((t.shape[3] - 1) // stride + 1)
            if conv.gemm_pays(pixels, cout, cin):
                return conv.conv1x1(t, w2, bias, residual, relu=relu, stride=stride)
            return conv.bias_act_(F.conv2d(t, w2.reshape(cout, cin, 1, 1), None, stride=stride), bias, residual, relu)

        def spatial(t, layer):
            w4, bias, stride, pad = layer
            cout, cin, r, _ = w4.shape
            pixels = (t.shape[0] * ((t.shape[2] + 2 * pad - r) // stride + 1)
                      * ((t.shape[3] + 2 * pad - r) // stride + 1))
            if cin % 64 == 0 and conv.gemm_pays(pixels, cout, cin * r * r):
                return conv.conv2d(t, w4, bias, None, relu=True, stride=stride, padding=pad)
            return conv.bias_act_(F.conv2d(t, w4, None, stride=stride, padding=pad), bias, None, True)

        stem, blocks = self._plan
        if self._stem_packed is not None and x.is_contiguous() and x.shape[3] % 2 == 0:
            # plain NCHW input (what clients send): the whole stem in one kernel
            y = conv.stem_conv_pool(x, *self._stem_packed)
        else:
            # Cin = 3 on the library kernel; its bias + ReLU ride in the pooling pass
            w4, bias, stride, pad = stem
            x = x.contiguous(memory_format=torch.channels_last)
            y = conv.bias_relu_maxpool(F.conv2d(x, w4, None, stride=stride, padding=pad), bias)
        main = torch.cuda.current_stream()
        side = self._side_stream if self.overlap_shortcut else None
        t_ahead = None  # the next block's first convolution, when the previous block's kernel already produced it
        for index, (a, mid, c, down) in enumerate(blocks):
            nxt = blocks[index + 1][0] if index + 1 < len(blocks) else None
            # projection shortcut inside the block kernel (first block of stage 1): the block input becomes a
            # second operand of the expansion GEMM; no separate kernel, no residual written and read back
            proj = self._proj.get(index) if (self.fused_expansion and y.shape[1] == 64 and _is_channels_last(torch, y)
                                             ) else None
            if proj is not None:
                identity = None
            elif down is None:
                identity = y
            elif side is None:
                identity = pointwise(y, down, None, False)
            else:
                # the projection shortcut only depends on the block input: run it on a second
                # stream (a parallel branch of the captured graph) next to the block's first two
                # convolutions, so it fills the tails those leave on the 256 CUs
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    identity = pointwise(y, down, None, False)
            t = t_ahead if t_ahead is not None else pointwise(y, a, None, True)
            t_ahead = None
            if down is not None and side is not None:
                main.wait_stream(side)
            attach = self._attach_next(mid, c, nxt)
            w4, bias, stride, pad = mid
            rows_w2 = self._rows_w2.get(index) if (self.fused_expansion and attach and t.shape[3] <= 62) else None
            if (rows_w2 is not None and (proj is not None or down is None) and (proj is None or nxt[0].shape[0] == 64)
                    and conv.tuned_rows(nxt[0].shape[0], proj is not None, t.shape[0])):
                # stage-1 block on the row-organised kernel: waves own pixel rows, no barrier behind the 3x3 part
                y, t_ahead = conv.bottleneck_rows(t, w4, bias, rows_w2, proj[1] if proj is not None else c[1],
                                                  None if proj is not None else identity, nxt[0], nxt[1],
                                                  proj_input=y if proj is not None else None)
                continue
            if proj is not None:
                n3_ok = attach and nxt[0].shape[0] == 64
                result = conv.conv2d_pointwise(t, w4, bias, proj[0], proj[1], None, relu=True, stride=1, padding=pad,
                                               proj_input=y, next_weight=nxt[0] if n3_ok else None,
                                               next_bias=nxt[1] if n3_ok else None)
                y, t_ahead = result if n3_ok else (result, None)
                continue
            fuse, tile_m = attach or self._fuse_expansion(t, mid, c), 0
            if self.fused_expansion and self._fusable(mid, c):
                # what was measured for exactly this block and batch (tools/tune_conv.py) overrides the rules
                mode = conv.tuned_block_mode(t.shape[2], t.shape[3], w4.shape[1], w4.shape[0], c[0].shape[0], stride,
                                             nxt[0].shape[0] if attach else 0, t.shape[0])
                if mode is not None:
                    fuse, tile_m = mode != "none", int(mode) if mode != "none" else 0
                    attach = attach and fuse
            if fuse:
                # 3x3 convolution and the 1x1 expansion in one kernel: the middle activation never
                # leaves LDS; with `attach` the kernel also runs the next block's first 1x1
                # convolution on the output chunks it has just produced
                if attach:
                    y, t_ahead = conv.conv2d_pointwise(t, w4, bias, c[0], c[1], identity, relu=True, stride=stride,
                                                       padding=pad, next_weight=nxt[0], next_bias=nxt[1], tile_m=tile_m)
                else:
                    y = conv.conv2d_pointwise(t, w4, bias, c[0], c[1], identity, relu=True, stride=stride, padding=pad,
                                              tile_m=tile_m)
            else:
                t = spatial(t, mid)
                y = pointwise(t, c, identity, True)
        return y

    @staticmethod
    def _fusable(mid, c):
        """Shapes the fused block kernel is instantiated for."""
        w4 = mid[0]
        return (w4.shape[0] in (64, 128) and w4.shape[1] % 64 == 0 and w4.shape[2] == 3 and c[2] == 1
                and c[0].shape[0] % 64 == 0)

    def _attach_next(self, mid, c, nxt):
        """Whether the fused block kernel also computes the next block's first 1x1 convolution
        (that convolution has stride 1 and 64 or 128 outputs; 128 with 128 middle channels).  It then
        never makes its own pass over the block output, the largest tensor of the stage:
        per transition 21 -> 17 us at batch 8, 47 -> 40 at batch 32, 171 -> 148 at batch 128."""
        if not self.fused_expansion or nxt is None:
            return False
        w4 = mid[0]
        n3 = nxt[0].shape[0]
        return (self._fusable(mid, c) and nxt[2] == 1 and nxt[0].shape[1] == c[0].shape[0]
                and (n3 in (64, 128) if w4.shape[0] == 64 else n3 == 128))

    def _fuse_expansion(self, t, mid, c):
        """Whether a block's 3x3 convolution and 1x1 expansion run as one kernel.  Measured on
        MI355X (``tools/dev/fused_block_bench.py``, us per block, two kernels -> fused): with 64
        middle channels 44 -> 34 at batch 32 and 170 -> 118 at batch 128 (the layers sit on the
        memory roofline and the fusion removes a write and a read of the middle activation),
        a tie at batch 8; with 128 middle channels 31 -> 29 at batch 32, 107 -> 90 and 84 -> 74
        at batch 128, slower below batch 32."""
        if not self.fused_expansion:
            return False
        w4, _, stride, pad = mid
        pixels = (t.shape[0] * ((t.shape[2] + 2 * pad - w4.shape[2]) // stride + 1)
                  * ((t.shape[3] + 2 * pad - w4.shape[3]) // stride + 1))
        if w4.shape[1] % 64 != 0 or c[2] != 1 or c[0].shape[0] % 64 != 0:
            return False
        if w4.shape[0] == 64:
            return pixels >= 16 * 3136
        return w4.shape[0] == 128 and pixels >= 32 * 784

    def _forward_into(self, x, out):
        torch = self._torch
        if self.fused:
            y = self._features_fused(x)  # takes the client's NCHW layout as it is
            if self.fused_tail:
                # global average pooling + fully connected layer + bias, fp32 logits straight
                # into `out` (two small kernels instead of reduction, GEMM and cast)
                from ..ops import conv

                conv.avgpool_linear(y, self._net.fc.weight, self._net.fc.bias, out)
                return
            feats = y.mean(dim=(2, 3))
        else:
            feats = self._net.features(x.contiguous(memory_format=torch.channels_last))
        logits = self._net.fc(feats)
        out.copy_(logits)  # fp16 -> fp32 cast fused into the copy kernel

    def run(self, x, out=None, instance=0, stable=False):
        """Forward ``x`` ([B,3,224,224] fp16 on this GPU).  When ``out`` (fp32
        [B,1000]) is given the logits are written there.  Work is enqueued on
        the stream of execution instance ``instance``; the caller synchronises
        via ``self.stream_of(instance)`` (``self.stream`` for instance 0).

        ``stable=True`` promises that ``x`` and ``out`` are long-lived buffers (a registered
        shared-memory region, the batcher's static buffers): a graph is captured for exactly
        these addresses and replayed on later calls.  Otherwise the tensors are copied through
        one pair of static buffers per (instance, batch) and that pair's graph is replayed, so
        callers with fresh tensors on every call neither capture nor retain anything per call."""
        torch = self._torch
        batch = x.shape[0]
        stream = self._streams[instance]
        if not self.use_graphs:
            if out is None:
                out = torch.empty(batch, 1000, device=self._device, dtype=torch.float32)
            with torch.no_grad(), torch.cuda.stream(stream):
                self._forward_into(x, out)
            return out
        if not (stable and out is not None):
            with self._lock:
                bufs = self._io_buffers.get((instance, batch))
                if bufs is None:
                    bufs = self._io_buffers[(instance, batch)] = (
                        torch.empty(batch, 3, 224, 224, device=self._device, dtype=torch.float16),
                        torch.empty(batch, 1000, device=self._device, dtype=torch.float32))
            sx, sout = bufs
            # the caller produced x on its own stream; results are handed back on the instance stream
            stream.wait_stream(torch.cuda.current_stream(self._device))
            with torch.cuda.stream(stream):
                sx.copy_(x, non_blocking=True)
            self._run_stable(sx, sout, instance)
            with torch.cuda.stream(stream):
                if out is None:
                    out = torch.empty(batch, 1000, device=self._device, dtype=torch.float32)
                    out.record_stream(stream)
                out.copy_(sout, non_blocking=True)
            return out
        return self._run_stable(x, out, instance)

    def _run_stable(self, x, out, instance):
        """Replay (capturing on first use) the graph of exactly these long-lived buffers."""
        torch = self._torch
        batch = x.shape[0]
        stream = self._streams[instance]
        key = (batch, x.data_ptr(), out.data_ptr(), instance)
        with self._lock:
            graph = self._graphs.get(key)
            if graph is None:
                graph = self._capture(x, out, stream, instance)
                self._graphs.put(key, graph, self._retire_event)
                if self.warmed_up:
                    self.online_captures += 1
            with torch.cuda.stream(stream), _trace.range(f"resnet50 graph replay b={batch}"):
                graph[0].replay()
        return out

    def _retire_event(self, key):
        event = self._torch.cuda.Event()
        event.record(self._streams[key[3]])
        return event

    def evict_region(self, base, nbytes):
        """Drop every captured graph that reads or writes ``[base, base + nbytes)`` (a
        shared-memory mapping that is about to be closed)."""
        end = base + nbytes
        with self._lock:
            self._graphs.evict_if(lambda k, v: base <= k[1] < end or base <= k[2] < end, self._retire_event)

    def graph_count(self):
        return len(self._graphs)

    def native_graph_plan(self, instances):
        """What the native executor needs to serve this model by itself: for every execution instance and
        every size in ``native_batches`` a captured graph over static buffers of its own, as
        ``(instance, batch, hipGraphExec_t, input address, output address)``, plus stream handles and the
        bytes per batch entry.  These graphs run on streams of their own and allocate their activations
        from pools of their own (one per instance): the dynamic batcher of the general path may replay ITS
        graphs on the model's streams at any moment from another thread, and two graph launches into one
        stream from two threads are not ordered as units - their kernels would interleave on shared
        activation memory.  ``keep`` holds graphs, buffers, streams and pools alive."""
        torch = self._torch
        graphs, keep = [], []
        sizes = [b for b in self.native_batches if b <= self.max_batch_size]
        if sizes[-1] != self.max_batch_size:
            sizes.append(self.max_batch_size)
        spec = self.inputs[0]
        in_dtype = torch.float16 if spec.datatype == "FP16" else torch.float32
        streams = [torch.cuda.Stream(device=self._device) for _ in range(instances)]
        pools = [torch.cuda.graph_pool_handle() for _ in range(instances)]
        for inst in range(instances):
            stream = streams[inst]
            for b in sizes:
                x = torch.zeros((b, *spec.dims), device=self._device, dtype=in_dtype)
                out = torch.empty((b, *self.outputs[0].dims), device=self._device, dtype=torch.float32)
                with torch.no_grad():
                    with torch.cuda.stream(stream):
                        self._forward_into(x, out)  # warm-up for this shape on this stream
                    stream.synchronize()
                    g = torch.cuda.CUDAGraph()
                    gate = self.capture_gate
                    if gate is not None:
                        gate.begin_capture()
                    try:
                        with torch.cuda.graph(g, pool=pools[inst], stream=stream):
                            self._forward_into(x, out)
                    finally:
                        if gate is not None:
                            gate.end_capture()
                graphs.append((inst, b, int(g.raw_cuda_graph_exec()), int(x.data_ptr()), int(out.data_ptr())))
                keep.append((g, x, out))
        keep.append((streams, pools))
        self._plan_streams = getattr(self, "_plan_streams", []) + streams  # unload() drops their workspaces
        return {"graphs": graphs, "keep": keep, "streams": [int(s.cuda_stream) for s in streams],
                "in_row_bytes": int(keep[0][1][0].numel() * keep[0][1].element_size()),
                "out_row_bytes": int(keep[0][2][0].numel() * 4), "max_batch": int(self.max_batch_size)}

    def _capture(self, x, out, stream, instance=0):
        """Capture one forward as a HIP graph.  All graphs of one execution instance allocate
        their activations from ONE memory pool: they only ever replay on that instance's
        stream, one after the other, and nothing inside the pool outlives a replay (the
        logits go to ``out``), so the footprint of an instance is that of its largest
        batch instead of the sum over every (batch, buffer) pair that was ever captured.
        Instances never share a pool - their graphs do run concurrently."""
        torch = self._torch
        with torch.no_grad():
            with torch.cuda.stream(stream):
                self._forward_into(x, out)  # warm-up for this shape
            stream.synchronize()
            pool = None  # A/B switch: a private pool per graph
            if os.environ.get("TCA_RESNET_SHARED_GRAPH_POOL", "1") != "0":
                if self._graph_pools[instance] is None:
                    self._graph_pools[instance] = torch.cuda.graph_pool_handle()
                pool = self._graph_pools[instance]
            g = torch.cuda.CUDAGraph()
            gate = self.capture_gate
            if gate is not None:
                gate.begin_capture()
            try:
                with torch.cuda.graph(g, pool=pool, stream=stream):
                    self._forward_into(x, out)
            finally:
                if gate is not None:
                    gate.end_capture()
        return (g, x, out)  # keep the tensors (hence the IPC mappings) alive

    @property
    def stream(self):
        return self._stream

    def stream_of(self, instance):
        return self._streams[instance]

    def execute(self, inputs, ctx):
        torch = self._torch
        x = inputs["input"]
        if x.dtype != torch.float16:
            x = x.to(torch.float16)
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, 224, 224):
            raise ModelError(f"unexpected shape for input 'input' for model 'resnet50'. Expected [-1,3,224,224], "
                             f"got {list(x.shape)}")
        out = inputs.get("__out__output")  # preallocated destination (e.g. a hipshm view)
        # the server only hands over long-lived buffers together with a destination: regions
        # registered by clients or the batcher's static buffers
        result = self.run(x, out, inputs.get("__instance__", 0), stable=out is not None)
        return {"output": result}


@register
class ResNet50Fp32In(ResNet50):
    """Same network behind an FP32 input (what stock ``image_client`` sends
    when the model metadata says FP32)."""

    name = "resnet50_fp32"
    native_kernel = None  # the input is converted on every call: no stable buffer for the native executor's graphs
    inputs = [TensorSpec("input", "FP32", [3, 224, 224], format="FORMAT_NCHW")]


@register
class PreprocessResNet50Ensemble(Model):
    """Encoded image files in, ImageNet logits out, everything after the decode on
    the GPU: JPEG / PNG bytes (a BYTES tensor, one file per batch entry) are decoded
    on the host (the C++ library's decoders, a request's files side by side on native threads),
    resized + INCEPTION-scaled + laid out as fp16 NCHW by the HIP preprocessing
    kernel (bit-exact with the Pillow pipeline of ``image_client``), and classified
    as a step through the loaded ``resnet50``'s dynamic batcher: concurrent ensemble
    requests are merged into one classifier batch like any other requests for that
    model.  This is the GPU counterpart of the reference's
    ``preprocess_inception_ensemble`` (server-side DALI preprocessing,
    ``src/python/examples/ensemble_image_client.py:60-110``)."""

    name = "preprocess_resnet50_ensemble"
    platform = "ensemble"
    device = "gpu"
    concurrent = True  # requests run side by side (decode pool, two preprocessors, the step's batcher merges them)
    max_batch_size = 64
    inputs = [TensorSpec("INPUT", "BYTES", [1])]
    outputs = [TensorSpec("OUTPUT", "FP32", [1000], label_filename="labels.txt")]

    def __init__(self):
        super().__init__()
        self.labels = [f"class_{i}" for i in range(1000)]
        self._pre = None
        self._torch = None
        self._device = None
        self._lock = threading.Lock()

    def load(self, device_index=None):
        import torch

        from ..ops.preprocess import Preprocessor

        self._torch = torch
        index = device_index or 0
        self._device = torch.device("cuda", index)
        # two preprocessors (staging buffers + stream each): one request stages its images while the kernel
        # of another runs
        self._pres = [Preprocessor(224, 224, 3, "NCHW", "FP16", scaling="INCEPTION", device=index,
                                   max_batch=self.max_batch_size) for _ in range(2)]
        self._pre = self._pres[0]
        self._pre_locks = [threading.Lock() for _ in self._pres]
        self._turn = 0
        self._decode_threads = int(os.environ.get("TCA_ENSEMBLE_DECODERS", "4"))  # per request, at most
        self._active = 0

    def unload(self):
        if self._pre is not None:
            for pre in self._pres:
                pre.close()
            self._pre = None

    def _classifier(self):
        net = _LOADED.get("resnet50")
        if net is None:
            raise ModelError("ensemble step 'resnet50' is not loaded on this server")
        return net

    @property
    def stream(self):
        return self._classifier().stream

    def execute(self, inputs, ctx):
        from ..ops.image_decode import decode_images

        torch = self._torch
        net = self._classifier()

        blobs = list(inputs["INPUT"].reshape(-1))
        with self._lock:
            self._active += 1
            # a lone request spreads its files over a few threads; many requests at once are parallel already
            threads = max(1, min(self._decode_threads, 8 // self._active))
        try:
            # one native call for the batch, no interpreter lock held meanwhile
            images = decode_images(blobs, channels=3, threads=threads)
        except ValueError as e:
            raise ModelError(f"unable to decode image: {e}") from None
        finally:
            with self._lock:
                self._active -= 1
        with self._lock:
            self._turn = which = (self._turn + 1) % len(self._pres)
        x = torch.empty(len(images), 3, 224, 224, device=self._device, dtype=torch.float16)
        with self._pre_locks[which]:  # one staging buffer set per preprocessor
            self._pres[which](images, x)  # synchronises its own stream: x is complete on return
        core = getattr(self, "core", None)
        if core is not None and getattr(net, "core", None) is core:
            return {"OUTPUT": core.infer_tensors("resnet50", {"input": x}, len(images))["output"]}
        with self._lock:  # stand-alone use (no serving core around the two models)
            out = net.run(x)
            net.stream.synchronize()
            return {"OUTPUT": out.float().cpu().numpy()}

    def config(self):
        cfg = super().config()
        cfg["ensemble_scheduling"] = {"step": [
            {"model_name": "image_preprocess_hip", "model_version": -1, "input_map": {"INPUT": "INPUT"},
             "output_map": {"OUTPUT": "preprocessed"}},
            {"model_name": "resnet50", "model_version": -1, "input_map": {"input": "preprocessed"},
             "output_map": {"output": "OUTPUT"}}]}
        return cfg
