"""A decode step captured as a chain of HIP graphs with the collectives launched between them.

For tensor parallelism the step is `segment, all-reduce, segment, all-reduce, ...` (two reductions per layer, one for a
sharded embedding, one all-gather for the head).  Capturing RCCL inside one big graph is possible but leaves the
communicator's kernels, its proxy thread and the capture rules entangled; a collective launched between two graph
replays is exactly the eager call the uncaptured step makes, and the host cost that matters — ~290 kernel launches per
step — collapses to one graph launch per segment.  The collectives work in place on buffers that belong to the graphs'
shared memory pool, so addresses stay fixed across replays.

`collective(fn)` is the only hook the layers need: eager mode runs `fn` at once; while a `SegmentedGraph` records it
closes the open segment, runs `fn` on the recording stream (so that communicators, workspaces and the allocator see
the same sequence as a replay) and opens the next segment."""
import contextlib
import gc
from typing import Callable, List, Optional, Union

import torch

_recording: Optional["SegmentedGraph"] = None

# Optional per-collective timing (bench.py --gpus N): event pairs on the launching stream around every collective that
# is issued eagerly — the uncaptured step and the seams of a segmented graph.  Collectives captured INSIDE a graph
# (`full` mode) cannot be bracketed; they show up in the step time only.
_timed: Optional[List] = None


def time_collectives(on: bool) -> None:
    global _timed
    _timed = [] if on else None


def collective_times_us() -> List[float]:
    """Durations of the collectives timed since time_collectives(True) (synchronises)."""
    if not _timed:
        return []
    torch.cuda.synchronize()
    return [a.elapsed_time(b) * 1e3 for a, b in _timed]


def _run_timed(fn: Callable[[], None]) -> None:
    if _timed is None or torch.cuda.is_current_stream_capturing():
        fn()
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    _timed.append((a, b))


@contextlib.contextmanager
def no_gc_during_capture():
    """The cyclic collector must not run while a stream is capturing: a finaliser it triggers (an older model's graphs
    and private pool, a retired workspace) calls into the HIP runtime, which invalidates the capture or aborts the
    process.  torch.cuda.graph() no longer collects on entry by default (torch >= 2.9), so collect here, then keep
    the collector off until the capture has ended."""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


def collective(fn: Callable[[], None]) -> None:
    if _recording is None:
        _run_timed(fn)
    else:
        _recording._seam(fn)


class SegmentedGraph:
    def __init__(self, device: torch.device, pool=None):
        self.device = device
        self.pool = pool if pool is not None else torch.cuda.graph_pool_handle()
        self.stream = torch.cuda.Stream(device)
        self.items: List[Union[torch.cuda.CUDAGraph, Callable[[], None]]] = []
        self._open: Optional[torch.cuda.CUDAGraph] = None

    # ---- recording ----------------------------------------------------------------------------------------------
    def _begin(self) -> None:
        g = torch.cuda.CUDAGraph()
        # thread_local: RCCL's proxy/watchdog threads may call the runtime while a segment is being captured
        g.capture_begin(pool=self.pool, capture_error_mode="thread_local")
        self._open = g

    def _end(self) -> None:
        g, self._open = self._open, None
        g.capture_end()
        self.items.append(g)

    def _seam(self, fn: Callable[[], None]) -> None:
        self._end()
        fn()
        self.items.append(fn)
        self._begin()

    def record(self, body: Callable[[], object]):
        """Run `body` once on the recording stream, splitting it at every `collective`; returns what `body` returned
        (tensors of the graphs' pool: valid after each `replay`)."""
        global _recording
        if _recording is not None:
            raise RuntimeError("nested SegmentedGraph.record")
        torch.cuda.synchronize(self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with no_gc_during_capture(), torch.cuda.stream(self.stream):
            _recording = self
            try:
                self._begin()
                out = body()
                self._end()
            except BaseException:
                if self._open is not None:
                    try:
                        self._open.capture_end()
                    except Exception:  # the capture was already invalidated
                        pass
                    self._open = None
                self.items.clear()
                raise
            finally:
                _recording = None
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return out

    # ---- replay -------------------------------------------------------------------------------------------------
    def replay(self) -> None:
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                _run_timed(it)

    @property
    def num_segments(self) -> int:
        return sum(isinstance(it, torch.cuda.CUDAGraph) for it in self.items)
