"""hipGraph capture of a fixed-geometry training step (torch.cuda.CUDAGraph == hipGraph on ROCm).

The engines already replay PREBUILT launch descriptors into STATIC buffers, so a whole step -- forward, the
hand-written backward, the fused optimizer launch, the few torch ops around them -- is a fixed sequence of kernel
launches on fixed addresses: exactly what a graph captures.  Where it pays: launch-bound steps.  The neural-filter
training step at the config's batch 2 issues ~80 launches for ~2 ms of GPU work and spends its time in host launch
gaps (reference src/ext_runner.py:39-76); the distillation step at batch 16 does not need it
(host enqueue 4.5 ms of a 107 ms step, bench.py `host_enqueue_ms_per_step`).

Rules for the captured function (checked by the caller, not here): same tensor shapes and the same tensor OBJECTS
every call (copy new data INTO them), no host synchronisation inside (`.item()`, `.cpu()`), host-side hyper-parameters
(learning rate, momentum ...) are baked in at capture time -- ``GraphedStep`` re-captures when ``key()`` changes.
"""
import torch


class GraphedStep(object):
    def __init__(self, fn, key=None, warmup=3):
        """fn(): one step on static tensors, returns a tensor (e.g. the loss) living in static memory;
        key(): hashable of everything host-side that the step bakes in (re-capture when it changes);
        warmup: eager calls before the first capture (lazy buffers, plans, cuBLAS-like handles must exist)."""
        self.fn, self.key, self.warmup = fn, key or (lambda: None), warmup
        self.calls, self.graph, self.out, self.graph_key = 0, None, None, None
        self.captures = 0

    def __call__(self):
        k = self.key()
        if self.calls < self.warmup:
            self.calls += 1
            return self.fn()
        if self.graph is None or k != self.graph_key:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.fn()
            self.graph, self.out, self.graph_key = graph, out, k
            self.captures += 1
        self.graph.replay()
        return self.out
