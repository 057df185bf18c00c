"""``GraphedBackward``: one training iteration's  loss = loss_fn(); loss.backward()  captured ONCE as a HIP graph and replayed.

The reference's training loops (src/neural_spectral/spectral_ode.py:178-189: zero_grad, forward, loss, backward, step) run the same
kernels on the same buffers every iteration; at BASELINE config 2 that is ~35 launches of 5-300 us, and what the HOST needs to enqueue
them is as long as what the device needs to run them.  Captured, the iteration is one graph launch (0.74 -> 0.56 ms on the same box,
profiles/r04_c2_graph.txt).  The library's kernels are launched on torch's current stream and allocate nothing, so they capture as
they are; the optimiser step stays outside the graph (Adam's bias corrections are host-side constants of the step count).

Contract: everything loss_fn reads besides the parameters must live in FIXED tensors (copy new data into them with ``copy_``); the
gradients are fixed buffers owned by the graph -- ``zero_grad`` is unnecessary (a replay overwrites them) and harmless
(``set_to_none=True`` is undone at the next call)."""
import torch


class GraphedBackward(object):
    def __init__(self, params, loss_fn, warmup=3):
        self.params = [p for p in params if p.requires_grad]
        if not self.params or not all(p.is_cuda for p in self.params):
            raise RuntimeError("GraphedBackward: parameters on the HIP device expected (the product has no CPU path)")
        # warm-up on a side stream (torch's capture rules): lazy initialisation inside the kernels' launchers (function attributes, workspaces,
        # the dense path's tables) must happen before the capture, not inside it
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                for p in self.params:
                    p.grad = None
                loss_fn().backward()
        torch.cuda.current_stream().wait_stream(side)
        for p in self.params:
            p.grad = None                                         # captured as "assign", not "accumulate"
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = loss_fn()
            self.loss.backward()
        self.grads = [p.grad for p in self.params]
        self.loss = self.loss.detach()

    def __call__(self):
        """Run the iteration: returns the loss (a fixed device scalar, overwritten by the next call); p.grad holds the new gradients."""
        self.graph.replay()
        for p, g in zip(self.params, self.grads):
            if p.grad is not g:
                p.grad = g
        return self.loss
