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

    def __init__(self, module: torch.nn.Module, warmup: int = 2, lane: int = 0, copy_inputs: bool = True):
        self.module = module
        self.warmup = warmup
        # copy_inputs=False: the graph is recorded on the caller's own input tensors (no copy-in per call); the caller keeps them alive
        # and refills them in place - calls with other tensors of the same shape are refused
        self.copy_inputs = copy_inputs
        self.lane = lane            # engine lane: its own scratch, the packed weights shared (PipelinedForward)
        self._graphs: Dict[Tuple, Tuple[torch.cuda.CUDAGraph, tuple, object]] = {}
        self._generation = -1

    @staticmethod
    def _key(args) -> Tuple:
        return tuple((tuple(a.shape), a.dtype, a.device.index) if isinstance(a, torch.Tensor) else a for a in args)

    def __call__(self, *args):
        from . import engine
        with engine.use_lane(self.lane):
            return self._call(engine, args)

    def _call(self, engine, args):
        key = self._key(args)
        if self._generation != engine.GENERATION[0]:
            self._graphs.clear()          # the packs these graphs point into are gone (or about to be rebuilt)
        entry = self._graphs.get(key)
        if entry is None:
            static_in = tuple(a.clone() if (isinstance(a, torch.Tensor) and self.copy_inputs) else a for a in args)
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
                if self.copy_inputs:
                    dst.copy_(src)
                elif dst.data_ptr() != src.data_ptr():
                    raise RuntimeError("egorear_amd.GraphedForward(copy_inputs=False): call with the tensors the graph was recorded on")
        graph.replay()
        return out


class PipelinedForward:
    """`lanes` captured forwards of one module replayed round-robin on `lanes` streams: consecutive batches overlap on the GPU, so the
    low-occupancy tail of one forward (fused transformer layers on 64-128 workgroups, sampling, small launches: about 1 of 11 ms at
    batch 64) runs under the convolutions of the next.  Throughput mode: a batch's outputs are static buffers of its lane, valid after
    `wait()` (or once the lane's stream has been waited on) and until the lane's next call.

        p = PipelinedForward(net)
        for img in batches: out = p(img)
        p.wait()
    """

    def __init__(self, module: torch.nn.Module, lanes: int = 2, warmup: int = 2, copy_inputs: bool = True):
        self.forwards = [GraphedForward(module, warmup, lane=i, copy_inputs=copy_inputs) for i in range(lanes)]
        self.streams = [torch.cuda.Stream() for _ in range(lanes)]
        self.k = 0

    def prime(self, *args):
        """Capture every lane for this input shape (outside any timed region)."""
        for _ in self.forwards:
            self(*args)
        self.wait()
        torch.cuda.synchronize()

    def __call__(self, *args):
        i = self.k % len(self.forwards)
        self.k += 1
        s = self.streams[i]
        s.wait_stream(torch.cuda.current_stream())          # the inputs were produced on the caller's stream
        if self.forwards[i].copy_inputs:
            # the copy into the lane's static input runs on the LANE's stream, up to two replays later: keep the caching allocator
            # from handing the caller's tensor to someone else before that copy has read it
            for a in args:
                if isinstance(a, torch.Tensor) and a.is_cuda:
                    a.record_stream(s)
        with torch.cuda.stream(s):
            return self.forwards[i](*args)

    def wait(self):
        cur = torch.cuda.current_stream()
        for s in self.streams:
            cur.wait_stream(s)
