"""Replay a calibrated quantised model as ONE HIP graph.

At small batch the quantised forward is launch-bound: every wrapper issues 2-3 short kernels (fake-quant of the
input, of the weight, the conv) and the host needs a few microseconds per launch.  Every entry point of the C
ABI is capturable (no allocation, no synchronisation, caller's stream) and the wrappers read no device scalar in
the steady state, so the whole forward can be captured once and replayed with a single launch.

    fwd = GraphedForward(model, example_input)     # model calibrated (one eager forward) and in eval mode
    y = fwd(x)                                      # x: same shape / dtype / memory format as the example

The returned tensor is the graph's static output buffer: clone it if it must survive the next call.
"""
import torch

__all__ = ["GraphedForward"]


class GraphedForward:
    def __init__(self, model, example, warmup=3):
        if not example.is_cuda:
            raise ValueError("GraphedForward needs a GPU example input")
        self.model = model
        self.static_in = example.clone(memory_format=torch.preserve_format)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):      # warm-up off the default stream (calibration, lazy inits)
            for _ in range(warmup):
                model(self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = model(self.static_in)

    def __call__(self, x):
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype:
            raise ValueError(f"graph captured for {tuple(self.static_in.shape)} {self.static_in.dtype}, "
                             f"got {tuple(x.shape)} {x.dtype}")
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out
