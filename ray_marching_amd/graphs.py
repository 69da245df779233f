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
    # The captured call's result is handed back DETACHED (same storage: replays keep updating it).  A result that still
    # carried its grad_fn would keep the autograd graph of the capture alive -- and with it AccumulateGrad nodes that
    # belong to the capture's side stream, which the next EAGER backward on another stream then meets (torch's
    # "AccumulateGrad node's stream does not match" warning, seen after the capture in round 2's GPU run).
    return graph, _detached(out), [p.grad for p in params]


def _detached(x):
    if isinstance(x, torch.Tensor):
        return x.detach()
    if isinstance(x, (tuple, list)):
        return type(x)(_detached(v) for v in x)
    if isinstance(x, dict):
        return {k: _detached(v) for k, v in x.items()}
    return x


class CapturedTrainingStep:
    """forward -> loss -> backward (-> optimiser) of a RenderLoop as ONE HIP graph, captured on the first call and
    replayed afterwards: the ~60 launches and the autograd bookkeeping of an eager step (40 AccumulateGrad nodes for the
    config-4 scene: 0.3-0.4 ms of host time against 0.40 ms of GPU work) become one graph launch plus two tiny pose copies.

        step = loop.training_step(lambda image: (image[..., :1] - target).pow(2).mean(), mode=0, marching_steps=64)
        for it in range(n):
            loss = step(orientations, translations)      # static tensor, overwritten by the next call
            optimiser.step()                              # or pass optimizer= to have it captured too (it must be capturable)

    The scene parameters are read in place by the kernels (CompiledScene.param_table), so optimiser steps and in-place
    edits between calls need nothing; `.grad` of every parameter is a static buffer the replays overwrite (set them to
    None / rebuild the step after changing which parameters require grad).  The pose may change from call to call (it
    is copied into static buffers); the frame size, shader mode, step count and the loss function are part of the graph.
    The first call runs two warm-up iterations at its pose before capturing (real iterations: with ``optimizer=`` they
    take optimiser steps, as torch's own whole-network capture recipe does)."""

    def __init__(self, loop, loss_fn, mode: int = 0, degree: int = 1, marching_steps: int = 32, optimizer=None,
                 pose_requires_grad: bool = False):
        self.loop, self.loss_fn, self.optimizer = loop, loss_fn, optimizer
        self.mode, self.degree, self.steps = int(mode), int(degree), int(marching_steps)
        self.pose_requires_grad = pose_requires_grad
        self.graph = None

    def _capture(self, orientations, translations):
        loop = self.loop
        self.q = orientations.detach().clone().requires_grad_(self.pose_requires_grad)
        self.t = translations.detach().clone().requires_grad_(self.pose_requires_grad)
        self.params = [p for p in loop.scene.parameters() if p.requires_grad]
        leaves = self.params + ([self.q, self.t] if self.pose_requires_grad else [])

        def step():
            loss = self.loss_fn(loop(self.q, self.t, self.mode, self.degree, self.steps))
            loss.backward()
            if self.optimizer is not None:
                self.optimizer.step()
            return loss

        self.graph, self.loss, self.grads = capture_step(step, leaves, warmup=2)

    def __call__(self, orientations, translations):
        if self.graph is None:
            self._capture(orientations, translations)
        with torch.no_grad():
            self.q.copy_(orientations)
            self.t.copy_(translations)
        self.graph.replay()
        return self.loss
