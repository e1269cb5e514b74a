"""hipGraph capture of the inference forward (serving mode).

The eval forward of lead-yolo-s is ~70 kernel launches of 5-200 us each.  Captured once into a hipGraph (torch.cuda.graph on
ROCm) it replays without host work, and the capture lets the model fork its small latency-bound branches (SE attention of
RFCBAMConv, the Detect heads of P3 / P4) onto an auxiliary stream at no cost (modules._overlap): -3 % step time at bs=32.
Every C-ABI entry point only launches kernels on the current stream (no allocation, no synchronisation), which is what makes
the whole forward capturable."""
import torch


class GraphedForward:
    """g = GraphedForward(model, example);  out = g(x)  replays the captured forward on x (same shape / dtype as example).
    The returned tensors are the graph's static outputs: consume or copy them before the next call."""

    def __init__(self, model, example, warmup=3):
        if model.training:
            raise RuntimeError("GraphedForward captures the inference forward: call model.eval() first")
        self.model = model
        self.x = example.clone()
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):           # warm up off the capture (weight packing caches, allocator)
            for _ in range(warmup):
                model(self.x)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = model(self.x)

    def __call__(self, x=None):
        if x is not None and x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x)
        self.graph.replay()
        return self.out
