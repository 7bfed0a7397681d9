"""hipGraph replay of a whole forward.

The hot path is ~150 kernel launches issued from Python through ctypes; at batch 64 the GPU is the bottleneck, but at
serving batch sizes the host launch cost (~10 us per launch) dominates.  Every C-ABI entry point only launches on the
caller's stream (no sync, no allocation), so the forward is capturable: `GraphedForward` captures it once per input
shape with torch.cuda.CUDAGraph (a hipGraph on ROCm) and replays it with one host call."""
from __future__ import annotations

from typing import Dict, Tuple

import torch


class GraphedForward:
    """Wrap an egorear_amd estimator (eval mode, weights on the device) for replayed inference.

        g = GraphedForward(net)
        preds, heatmaps = g(img)            # first call per shape: warm-up + capture; later calls: copy-in + replay

    Outputs are static buffers owned by the graph: they are overwritten by the next call (clone what must be kept).

    A graph holds raw pointers to the packed weights it was recorded with.  Whenever packs are dropped or rebuilt
    (`load_state_dict`, `engine.invalidate`, a training forward: all bump `engine.GENERATION`) the recorded graphs are
    discarded and the next call captures again - a replay never reads freed or stale weights."""

    def __init__(self, module: torch.nn.Module, warmup: int = 2):
        self.module = module
        self.warmup = warmup
        self._graphs: Dict[Tuple, Tuple[torch.cuda.CUDAGraph, tuple, object]] = {}
        self._generation = -1

    @staticmethod
    def _key(args) -> Tuple:
        return tuple((tuple(a.shape), a.dtype, a.device.index) if isinstance(a, torch.Tensor) else a for a in args)

    def __call__(self, *args):
        from . import engine
        key = self._key(args)
        if self._generation != engine.GENERATION[0]:
            self._graphs.clear()          # the packs these graphs point into are gone (or about to be rebuilt)
        entry = self._graphs.get(key)
        if entry is None:
            static_in = tuple(a.clone() if isinstance(a, torch.Tensor) else a for a in args)
            with torch.no_grad():
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):      # warm-up off the default stream: packs weights, primes the allocator
                    for _ in range(self.warmup):
                        self.module(*static_in)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = self.module(*static_in)
            entry = (graph, static_in, out)
            self._graphs[key] = entry
            self._generation = engine.GENERATION[0]   # after warm-up + capture: the packs exist now
        graph, static_in, out = entry
        for dst, src in zip(static_in, args):
            if isinstance(dst, torch.Tensor):
                dst.copy_(src)
        graph.replay()
        return out
