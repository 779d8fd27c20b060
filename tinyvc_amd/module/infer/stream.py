"""Real-time wrapper: rolling input buffer, full convert per block, SOLA alignment and cross-fade
(reference module/infer/stream.py:30-96).  `StreamInfer` is the reference's single-stream class;
`BatchedStreamInfer` runs S independent streams through one batched convert + one SOLA launch with
the lag arg-max kept on the device."""
import numpy as np
import torch

from .generator import Generator


def _fade_windows(crossfade_size, device):
    # stream.py:61-62, computed on the host exactly as the reference does, then uploaded
    fade_in = torch.sin(np.pi * torch.arange(0, 1, 1 / crossfade_size) / 2) ** 2
    return fade_in.to(device), (1 - fade_in).to(device)


class BatchedStreamInfer:
    def __init__(self, generator: Generator, n_streams=1, target=None, pitch_shift=0., device=None,
                 block_size=1920, extra_size=0, use_phase_vocoder=False, f0_estimation="default", use_graph=False):
        self.generator = generator
        self.n_streams = n_streams
        self.target = target
        self.pitch_shift = pitch_shift
        self.device = torch.device(device) if device is not None else generator._module_device()
        self.block_size = block_size
        self.extra_size = extra_size
        self.sola_search_size = 1920
        self.last_dilay_size = 3840          # (sic) the reference's attribute name, stream.py:49
        self.crossfade_size = 1920
        self.use_phase_vocoder = use_phase_vocoder
        self.f0_estimation = f0_estimation
        self.input_size = max(self.block_size + self.crossfade_size + self.sola_search_size + 2 * self.last_dilay_size,
                              self.block_size + self.extra_size)
        self.last_shift = None
        # use_graph: after two eager warm-up blocks the whole per-block pipeline (buffer roll, noise draw,
        # convert, SOLA) is captured once into a HIP graph and replayed per block: ~170 launches become one.
        self.use_graph = use_graph
        self._graph = None
        self._calls = 0

    def init_buffer(self):
        self.fade_in_window, self.fade_out_window = _fade_windows(self.crossfade_size, self.device)
        self.input_wav = torch.zeros(self.n_streams, self.input_size, device=self.device)
        self.sola_buffer = torch.zeros(self.n_streams, self.crossfade_size, device=self.device)
        self._graph = None
        self._calls = 0

    def _step(self, blocks, noise_angle):
        # torch.roll + slice assignment of the reference (stream.py:69-70), in place on a fixed buffer, one launch
        self.generator.engine(self.device).stream_push(self.input_wav, blocks)
        if noise_angle is None and self.use_graph:
            # a captured step bakes kernel arguments in: the library's seeded draw would replay ONE seed for every block.  torch's
            # generator is graph-safe (its offset advances per replay), so the graph path keeps the reference's torch.rand draw
            from ..tinyvc import Decoder
            noise_angle = Decoder.draw_noise_angle(self.n_streams, self.input_size // 480, self.device)
        y = self.generator.convert(self.input_wav, self.target, self.pitch_shift, device=self.device,
                                   f0_estimation=self.f0_estimation, noise_angle=noise_angle)
        eng = self.generator.engine(self.device)
        return eng.sola(y, self.sola_buffer, self.fade_in_window, self.block_size,
                        self.use_phase_vocoder, want_shift=True)

    def _graph_key(self):
        """Everything a captured step bakes in as raw device pointers: the packed weights (re-packed - and the old arena
        freed - when a parameter changes), the engine's scratch workspace (re-allocated when another call needs a bigger
        one), the prepared index riding on `target`, plus the scalars captured by value."""
        eng = self.generator.engine(self.device)          # re-packs the weights first if a parameter changed
        ws, tgt = eng._ws, self.target
        return (eng.weights_key, ws.data_ptr() if ws is not None else 0, ws.numel() if ws is not None else 0,
                id(tgt), tgt._version, tgt.data_ptr(), float(self.pitch_shift), bool(self.use_phase_vocoder))

    @torch.no_grad()
    def audio_callback(self, blocks, noise_angle=None):
        """blocks [S, block_size] -> converted blocks [S, block_size]."""
        blocks = blocks.to(self.device)
        self._calls += 1
        if not self.use_graph or self._calls <= 2:
            out, shift = self._step(blocks, noise_angle)
            self.last_shift = shift
            return out
        key = self._graph_key()
        if self._graph is not None and key != self._graph[0]:
            self._graph = None        # stale pointers inside the captured graphs: drop them and capture again
        if self._graph is None:
            self._graph = (key, {})
            self._g_in = torch.zeros(self.n_streams, self.block_size, device=self.device)
            self._g_angle = torch.zeros(self.n_streams, 961, self.input_size // 480, device=self.device)
        inject = noise_angle is not None      # injected phases replay from a static buffer; otherwise the draw is part of the graph
        graphs = self._graph[1]
        if inject not in graphs:
            self._g_in.copy_(blocks)
            if inject:
                self._g_angle.copy_(noise_angle)
            # capture changes no state: the buffer roll, convert and SOLA are recorded, not run
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                res = self._step(self._g_in, self._g_angle if inject else None)
            graphs[inject] = (g, res)
            if self._graph_key() != key:      # the capture itself grew the workspace: what it recorded is already stale
                self._graph = None
                out, shift = self._step(blocks, noise_angle)
                self.last_shift = shift
                return out
        self._g_in.copy_(blocks)
        if inject:
            self._g_angle.copy_(noise_angle)
        g, (g_out, g_shift) = graphs[inject]
        g.replay()
        self.last_shift = g_shift
        return g_out.clone()


class StreamInfer(BatchedStreamInfer):
    """Single stream with the reference's constructor and 1-D buffers (stream.py:31-57)."""

    def __init__(self, generator: Generator, target=None, pitch_shift=0., device=torch.device("cpu"),
                 block_size=1920, extra_size=0, use_phase_vocoder=False, f0_estimation="default", use_graph=False):
        dev = torch.device(device)
        if dev.type != "cuda":
            dev = generator._module_device()     # the reference defaults to CPU; there is no CPU path here
        super().__init__(generator, 1, target, pitch_shift, dev, block_size, extra_size, use_phase_vocoder, f0_estimation, use_graph)

    @torch.no_grad()
    def audio_callback(self, block, noise_angle=None):
        """block [block_size] -> converted block [block_size] (stream.py:68-96)."""
        return super().audio_callback(block[None], noise_angle)[0]
