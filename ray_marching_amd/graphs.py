"""Capturing a whole training step (forward + backward [+ optimiser]) of the fused frame in a HIP graph.

The recipe that matters (found the hard way): warm-up iterations and the capture must run
on THE SAME side stream.  Autograd's AccumulateGrad node of a parameter remembers the stream it was created on
and outlives the iteration that made it; ``torch.cuda.graph()`` by default captures on a fresh stream of its own,
so the first backward under capture finds AccumulateGrad nodes that belong to the warm-up stream, and the engine
inserts an event wait between a capturing and a non-capturing stream -- torch warns ("The AccumulateGrad node's
stream does not match ..."), and depending on what else is in flight the capture is invalidated
(hipErrorStreamCaptureInvalidated; under rocprofv3 the process aborted).  Passing the warm-up stream to
``torch.cuda.graph(stream=...)`` removes the cross-stream edge altogether.
"""
from __future__ import annotations

import torch


def capture_step(step, params, warmup: int = 3, before_each=None):
    """Run ``step()`` ``warmup`` times on a side stream, then capture one more call of it on the same stream.

    ``params``: the tensors whose ``.grad`` the step produces (set to None before every warm-up call and before
    the capture, so the captured backward allocates the gradient buffers inside the graph's pool).
    ``before_each``: optional callable run before every warm-up step (e.g. ``optimizer.zero_grad``).
    Returns ``(graph, result_of_the_captured_call, [p.grad for p in params])`` -- the gradient tensors the
    replays write to."""
    params = list(params)
    dev = params[0].device if params else torch.device("cuda", torch.cuda.current_device())
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(warmup):
            for p in params:
                p.grad = None
            if before_each is not None:
                before_each()
            step()
    torch.cuda.current_stream(dev).wait_stream(side)
    for p in params:
        p.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        out = step()
    return graph, out, [p.grad for p in params]
