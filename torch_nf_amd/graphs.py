"""HIP-graph capture of a whole training step.

The flows this package serves are small (the reference's LFI scripts: D = 6, a few thousand contexts), so a
step of  loss -> backward -> Adam  is some forty short launches and the host, not the GPU, sets the pace
(1.4 ms per step in eager mode for `scripts/lfi_mat.py`'s configuration, 0.6 ms replayed as one graph on an
MI355X).  `GraphedStep` captures one call of a step function into a HIP graph and replays it.

Rules for the step function (the usual ones of graph capture):
  * it reads its inputs from tensors that stay at the same address -- refresh them with `copy_()`;
    device-side random draws inside the step (`torch.randint`, `torch.randn` on the device) are fine;
  * no host synchronisation inside (`.item()`, `.cpu()`, printing a loss);
  * optimisers built with `capturable=True`;
  * it returns tensors (or nothing); `GraphedStep.__call__` returns the same tensor objects, refreshed.
"""
import torch


class GraphedStep:
    def __init__(self, fn, warmup=3):
        """Runs `fn` `warmup` times on a side stream (they are real steps), then captures one more call."""
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a HIP device")
        self.warmup_outputs = []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                out = fn()
                self.warmup_outputs.append(_detached_copy(out))
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn()

    def __call__(self):
        self.graph.replay()
        return self.outputs


def _detached_copy(out):
    if out is None:
        return None
    if torch.is_tensor(out):
        return out.detach().clone()
    return type(out)(_detached_copy(o) for o in out)
