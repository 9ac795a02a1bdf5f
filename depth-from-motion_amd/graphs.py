"""hipGraph capture of launch-bound sub-graphs (inference).

The 2-D necks either side of the path (SPPUNetNeck, BEVHourglass) are ~100 small
kernels each -- window means, 1x1 convolutions on a few hundred pixels, GroupNorm passes, bilinear
up-samplings, the MFMA 3x3 convolutions -- whose host-side launch cost (~10 us each through Python) is
longer than their device time: 0.98 ms of wall clock for ~0.5 ms of device work per SPPUNetNeck
forward at config K (profiles/archive/r02_c62_neck2d_timing.txt).  ``GraphedCallable`` records such a callable
once per input signature into a hipGraph (``torch.cuda.CUDAGraph`` is hipGraph on ROCm; the package's
own kernels launch on torch's current stream, which is the capture stream) and replays it with one
launch.  Only tensor -> tensor callables without host synchronisation qualify: no ``.item()``, no
host uploads inside (a captured copy from pinned host memory would re-read whatever that buffer holds
at replay time), no autograd."""
import torch


def _flatten(out):
    if torch.is_tensor(out):
        return [out], lambda ts: ts[0]
    if isinstance(out, (tuple, list)):
        parts = [_flatten(o) for o in out]
        sizes = [len(p[0]) for p in parts]
        flat = [t for p in parts for t in p[0]]

        def rebuild(ts, parts=parts, sizes=sizes, kind=type(out)):
            res, i = [], 0
            for (_, rb), n in zip(parts, sizes):
                res.append(rb(ts[i:i + n]))
                i += n
            return kind(res)
        return flat, rebuild
    if out is None:
        return [], lambda ts: None
    raise TypeError(f'GraphedCallable: unsupported output type {type(out)}')


class GraphedCallable:
    """``g = GraphedCallable(fn); y = g(tensors)`` with ``fn(list_of_tensors) -> tensor | tuple | list``.

    The first call with a new signature (shapes, dtypes, strides, device) warms ``fn`` up on a side
    stream (lazy initialisation, MIOpen find, workspace growth all happen there), captures one call
    into a graph that reads from static input buffers, and replays it; later calls copy their inputs
    into those buffers and replay.  The returned tensors are the graph's static outputs: valid until the
    next call with the same signature (``clone()`` what must live longer).  Falls through to a plain
    call when autograd is recording or an input is not on the GPU."""

    def __init__(self, fn, warmup=3, max_graphs=8):
        self.fn, self.warmup, self.max_graphs = fn, warmup, max_graphs
        self._graphs = {}

    @staticmethod
    def _key(tensors):
        return tuple((tuple(t.shape), t.dtype, tuple(t.stride()), str(t.device)) for t in tensors)

    def __call__(self, tensors):
        tensors = list(tensors)
        if torch.is_grad_enabled() or not tensors or not all(t.is_cuda for t in tensors):
            return self.fn(tensors)
        key = self._key(tensors)
        entry = self._graphs.get(key)
        if entry is None:
            if len(self._graphs) >= self.max_graphs:
                return self.fn(tensors)
            entry = self._graphs[key] = self._capture(tensors)
        static_in, graph, static_out, rebuild = entry
        for s, t in zip(static_in, tensors):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t)
        graph.replay()
        return rebuild(static_out)

    def _capture(self, tensors):
        dev = tensors[0].device
        # same strides as the caller's tensors (channels_last inputs stay channels_last)
        static_in = [torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev).copy_(t) for t in tensors]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self.fn(static_in)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.fn(static_in)
        static_out, rebuild = _flatten(out)
        return static_in, graph, static_out, rebuild
