"""Clip sharding across the GPUs of one node.

Clips are independent (no cross-clip op exists on the path), so a batch is cut into contiguous
blocks, one per rank; weights are replicated; nothing is exchanged while encoding/decoding.  The
only collective is the end-of-step gather: codes to every rank (8*L bytes per clip) and, when one
rank wants the audio, waveforms to that rank.  One process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


def shard_bounds(n_clips: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) block of clips for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_clips, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _pad_rows(x: torch.Tensor, dim: int, n: int) -> torch.Tensor:
    if x.shape[dim] == n:
        return x.contiguous()
    shape = list(x.shape)
    shape[dim] = n - x.shape[dim]
    return torch.cat([x, x.new_zeros(shape)], dim=dim).contiguous()


def gather_codes(codes_local: torch.Tensor, dist, world: int, counts: Optional[List[int]] = None) -> torch.Tensor:
    """codes_local (K, b_r, L) -> (K, sum b_r, L) on every rank, ranks in order.  Collectives need equal
    sizes, so uneven shards are zero-padded to the largest and trimmed after the gather."""
    if world == 1:
        return codes_local
    K, b, L = codes_local.shape
    counts = counts or [b] * world
    bmax = max(counts)
    out = torch.empty((world * K, bmax, L), dtype=codes_local.dtype, device=codes_local.device)
    dist.all_gather_into_tensor(out, _pad_rows(codes_local, 1, bmax))     # rank-major concatenation on dim 0
    out = out.view(world, K, bmax, L)
    return torch.cat([out[r, :, :counts[r]] for r in range(world)], dim=1)


def gather_waveforms(wav_local: torch.Tensor, dist, world: int, rank: int, dst: int = 0,
                     counts: Optional[List[int]] = None) -> Optional[torch.Tensor]:
    """wav_local (b_r, T) -> (sum b_r, T) on rank `dst`, None elsewhere."""
    if world == 1:
        return wav_local
    b, T = wav_local.shape
    counts = counts or [b] * world
    bmax = max(counts)
    bufs = [torch.empty((bmax, T), dtype=wav_local.dtype, device=wav_local.device) for _ in range(world)] if rank == dst else None
    dist.gather(_pad_rows(wav_local, 0, bmax), bufs, dst=dst)
    return torch.cat([bufs[r][:counts[r]] for r in range(world)], dim=0) if rank == dst else None


class PendingGather:
    """An end-of-step exchange in flight on the collective stream (RCCL runs it beside the next step's kernels).
    `result()` waits for it (stream-wise on GPU backends) and assembles (codes of all clips, waveforms on `dst`)."""

    def __init__(self, works, codes_buf, wav_big, wav_bufs, counts, world, K, rank, dst):
        self._works, self._codes_buf, self._wav_big, self._wav_bufs = works, codes_buf, wav_big, wav_bufs
        self._counts, self._world, self._K, self._rank, self._dst = counts, world, K, rank, dst

    def result(self):
        for w in self._works:
            w.wait()
        bmax = max(self._counts)
        even = min(self._counts) == bmax
        c = self._codes_buf.view(self._world, self._K, bmax, -1)
        if even and self._K == 1:
            codes = c.view(1, self._world * bmax, -1)                # ranks are already in clip order: no copy
        else:
            codes = torch.cat([c[r, :, :self._counts[r]] for r in range(self._world)], dim=1)
        wav = None
        if self._rank == self._dst:
            if even:
                wav = self._wav_big.view(self._world * bmax, -1)     # gathered straight into one buffer: no copy
            else:
                wav = torch.cat([self._wav_bufs[r][:self._counts[r]] for r in range(self._world)], dim=0)
        return codes, wav


def gather_async(codes_local: torch.Tensor, wav_local: torch.Tensor, dist, world: int, rank: int, dst: int = 0,
                 counts: Optional[List[int]] = None) -> "PendingGather":
    """Non-blocking form of gather_codes + gather_waveforms: both collectives are issued with async_op=True, so on
    RCCL they overlap the kernels the caller enqueues next; the buffers stay referenced by the returned object."""
    K, b, L = codes_local.shape
    T = wav_local.shape[1]
    counts = counts or [b] * world
    bmax = max(counts)
    codes_buf = torch.empty((world * K, bmax, L), dtype=codes_local.dtype, device=codes_local.device)
    w1 = dist.all_gather_into_tensor(codes_buf, _pad_rows(codes_local, 1, bmax), async_op=True)
    # the per-rank receive buffers are slices of ONE tensor, so with even shards the gathered batch needs no concatenation
    wav_big = torch.empty((world, bmax, T), dtype=wav_local.dtype, device=wav_local.device) if rank == dst else None
    wav_bufs = list(wav_big.unbind(0)) if rank == dst else None
    w2 = dist.gather(_pad_rows(wav_local, 0, bmax), wav_bufs, dst=dst, async_op=True)
    return PendingGather([w1, w2], codes_buf, wav_big, wav_bufs, counts, world, K, rank, dst)


def roundtrip_sharded(model, wav_all: torch.Tensor, bandwidth_id: torch.Tensor, dist, rank: int, world: int,
                      dst: int = 0):
    """encode_infer + decode of this rank's block of `wav_all` (B, T); returns (codes of all clips on every
    rank, waveforms of all clips on rank dst)."""
    lo, hi = shard_bounds(wav_all.shape[0], rank, world)
    counts = [shard_bounds(wav_all.shape[0], r, world)[1] - shard_bounds(wav_all.shape[0], r, world)[0] for r in range(world)]
    feats, codes = model.encode_infer(wav_all[lo:hi].contiguous(), bandwidth_id=bandwidth_id)
    out = model.decode(feats, bandwidth_id=bandwidth_id)
    return gather_codes(codes, dist, world, counts), gather_waveforms(out, dist, world, rank, dst, counts)


class StepRunner:
    """encode_infer + decode steps of one (model, batch) with the end-of-step exchange of the sharded mode (bench.py
    --gpus N drives this; the world-2 tests drive it with a recording stub in place of the model).

    Ordering invariant: the exchange of step i (codes to every rank, 8*L bytes per clip; waveforms to
    rank `dst`) is ISSUED right after step i+1's encode_infer has been enqueued and COLLECTED before step i+1 returns.
    On RCCL the collective's stream waits for what was enqueued before it, so its kernels start when that encode has
    finished on the GPU, run beside step i+1's decode, and the next encode is ordered behind them by the collecting
    stream wait: RCCL never runs beside lstm_persist_kernel, which needs every CU for its resident workgroups
    (wavtokenizer_amd/csrc/lstm_persist.hip).  `log`, when given, receives (event, step) tuples in host order:
    "encode", "issue", "decode", "collect" - the tests assert the invariant on it."""

    def __init__(self, model, wav, bw, dist, world, rank, gather=True, backend="nccl", dst=0, log=None, lanes=1,
                 force_collectives=False):
        self.model, self.wav, self.bw = model, wav, bw
        self.dist, self.world, self.rank, self.backend, self.dst = dist, world, rank, backend, dst
        # force_collectives: run the exchange in a world of ONE rank too (bench.py --force-collectives, the rccl_world1 row and
        # test): the RCCL calls, their stream semantics and their interplay with the persistent LSTM are then exercised on the
        # single GPU a development box has
        self.gather = bool(gather) and (world > 1 or (force_collectives and dist is not None))
        # lanes > 1 (only without the exchange): step i runs on HIP stream i % lanes, so step i+1's encode_infer runs beside
        # step i's decode and fills the launch gaps and last-round tails of its kernels.  The model keeps a plan + workspace
        # per stream (pretrained._Engine._key) and the library chains the persistent LSTM launches of different streams
        # (capi.cpp, LstmChain): at most one of them is on the GPU at a time.  Same kernels, same results.
        if lanes > 1 and self.gather:
            raise ValueError("lanes > 1 is the single-rank pipeline; the sharded exchange keeps one lane (DESIGN section 6)")
        self.lanes = int(lanes)
        self.streams = [torch.cuda.Stream(device=wav.device) for _ in range(self.lanes)] if self.lanes > 1 else None
        self.prev = None            # (step index, codes, waveform) of the previous step, not yet exchanged
        self.i = 0
        self.log = log
        self.wait_s = 0.0           # host time spent collecting exchanges (PendingGather.result)
        self.exchanges = 0

    def _note(self, what, step):
        if self.log is not None:
            self.log.append((what, step))

    def _issue(self):
        step, codes, out = self.prev
        self.prev = None
        if self.backend == "gloo":                               # rehearsal only: through host memory
            codes, out = codes.cpu(), out.cpu()
        self._note("issue", step)
        return step, gather_async(codes, out, self.dist, self.world, self.rank, dst=self.dst)

    def _collect(self, pend):
        import time
        step, p = pend
        t0 = time.perf_counter()
        res = p.result()
        self.wait_s += time.perf_counter() - t0
        self.exchanges += 1
        self._note("collect", step)
        return res

    def step(self):
        if self.streams is not None:
            st = self.streams[self.i % self.lanes]
            self.i += 1
            # the lane starts behind whatever the caller's stream holds (a waveform the caller is still producing there);
            # the results are ordered with the caller's stream only by drain(): do not read them before
            st.wait_stream(torch.cuda.current_stream(self.wav.device))
            with torch.cuda.stream(st):
                feats, codes = self.model.encode_infer(self.wav, bandwidth_id=self.bw)
                out = self.model.decode(feats, bandwidth_id=self.bw)
            return codes, out, None
        i = self.i
        self.i += 1
        feats, codes = self.model.encode_infer(self.wav, bandwidth_id=self.bw)
        self._note("encode", i)
        pend = self._issue() if (self.gather and self.prev is not None) else None
        out = self.model.decode(feats, bandwidth_id=self.bw)
        self._note("decode", i)
        res = self._collect(pend) if pend is not None else None
        if self.gather:
            self.prev = (i, codes, out)
        return codes, out, res

    def drain(self):
        """The exchange of the last step: issued and collected here (inside the timed region of bench.py).  With lanes the
        caller's stream is made to wait for every lane (bench.py synchronises the device right after)."""
        if self.streams is not None:
            cur = torch.cuda.current_stream(self.wav.device)
            for st in self.streams:
                cur.wait_stream(st)
            return None
        if self.gather and self.prev is not None:
            return self._collect(self._issue())
        return None


def check_exchange_order(log, n_steps):
    """Asserts DESIGN section 6's invariant on a StepRunner log of n_steps steps followed by drain()."""
    pos = {ev: k for k, ev in enumerate(log)}
    assert len(pos) == len(log), "an event was logged twice"
    for i in range(n_steps):
        assert pos[("encode", i)] < pos[("decode", i)]
        if i + 1 < n_steps:
            # issued after the NEXT encode is enqueued, before the next decode; collected before the step after that begins
            assert pos[("encode", i + 1)] < pos[("issue", i)] < pos[("decode", i + 1)] < pos[("collect", i)]
            if i + 2 < n_steps:
                assert pos[("collect", i)] < pos[("encode", i + 2)]
        else:
            assert pos[("decode", i)] < pos[("issue", i)] < pos[("collect", i)]      # the drain


class HostPipeline:
    """The round trip as the reference's own script runs it (infer.py:44-70: waveform in host memory -> encode_infer ->
    decode -> 16-bit PCM back in host memory), for batches, with the host <-> device legs overlapped with compute.  Step i
    runs on lane i % lanes, each lane a HIP stream that does, in order,

        H2D   pinned host waveforms (B, T) fp32 -> device                      (18.4 MB at 64 x 3 s)
        encode_infer + decode + the PCM16 conversion (wt_pcm16)
        D2H   int16 samples -> pinned host buffer                              (9.2 MB: half the fp32 bytes)

    so the copies of one lane run beside the kernels of the other and no event crosses streams (a first version with
    separate copy streams and cross-stream events per step lost 12 % to the dependency packets: 5.74 vs 5.08 ms per step).
    A lane's device and host buffers are reused every `lanes` steps; stream order alone makes that safe.  The copy legs are
    timed on their own (`copy_times_ms`: the same copies on a side stream, without dependencies, while the lanes compute):
    SURVEY 8(d) asks for H2D separately."""

    def __init__(self, model, B: int, T: int, bw, lanes: int = 2, device=None, limit: float = 0.99):
        import ctypes
        from . import _capi
        self._ctypes, self._capi = ctypes, _capi
        self.model, self.bw, self.B, self.T, self.lanes, self.limit = model, bw, B, T, int(lanes), float(limit)
        dev = torch.device(device) if device is not None else model._device()
        self.dev = dev
        L = model.arch.frames(T)
        self.n_out = model._wave_len(L)
        self.h_in = [torch.empty((B, T), dtype=torch.float32).pin_memory() for _ in range(self.lanes)]
        self.h_out = [torch.empty((B, self.n_out), dtype=torch.int16).pin_memory() for _ in range(self.lanes)]
        self.d_in = [torch.empty((B, T), dtype=torch.float32, device=dev) for _ in range(self.lanes)]
        self.d_pcm = [torch.empty((B, self.n_out), dtype=torch.int16, device=dev) for _ in range(self.lanes)]
        self.d_ws = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(self.lanes)]
        self.s_lane = [torch.cuda.Stream(device=dev) for _ in range(self.lanes)]
        self.i = 0
        self.codes = [None] * self.lanes

    def step(self):
        ct, capi = self._ctypes, self._capi
        k = self.i % self.lanes
        self.i += 1
        st = self.s_lane[k]
        with torch.cuda.stream(st):
            self.d_in[k].copy_(self.h_in[k], non_blocking=True)
            feats, codes = self.model.encode_infer(self.d_in[k], bandwidth_id=self.bw)
            out = self.model.decode(feats, bandwidth_id=self.bw)
            capi.check(capi.lib.wt_pcm16(ct.c_void_p(out.data_ptr()), out.numel(), self.limit, 0, ct.c_void_p(self.d_pcm[k].data_ptr()),
                                         ct.c_void_p(self.d_ws[k].data_ptr()), ct.c_void_p(st.cuda_stream)), "wt_pcm16")
            self.h_out[k].copy_(self.d_pcm[k], non_blocking=True)
            self.codes[k] = codes
        return k

    def drain(self):
        """Everything submitted so far is in host memory when this returns."""
        for s in self.s_lane:
            s.synchronize()

    def copy_times_ms(self, steps: int = 8, busy=None):
        """(h2d ms, d2h ms) of one step's copies, measured on a side stream beside compute (median of `steps`).  `busy()`, when
        given, enqueues one resident-input round trip on the caller's stream per measurement: the copies are then timed beside
        the codec's kernels but with the copy engines to themselves.  Without it the lanes' own steps run meanwhile - and
        their copies share the engines' queues with the timed ones in submission order, a lane's H2D waiting (in stream order)
        for that lane's previous step: the timed copy then reads as the wait in front of it (9 ms for a 0.34 ms copy)."""
        self.drain()
        side = torch.cuda.Stream(device=self.dev)
        d_in = torch.empty_like(self.d_in[0])
        h_out = torch.empty_like(self.h_out[0]).pin_memory()
        h2d, d2h = [], []
        for _ in range(steps):
            if busy is not None:
                busy()
            else:
                self.step()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            with torch.cuda.stream(side):
                ev[0].record()
                d_in.copy_(self.h_in[0], non_blocking=True)
                ev[1].record()
                ev[2].record()
                h_out.copy_(self.d_pcm[0], non_blocking=True)
                ev[3].record()
            ev[3].synchronize()
            h2d.append(ev[0].elapsed_time(ev[1]))
            d2h.append(ev[2].elapsed_time(ev[3]))
        self.drain()
        h2d.sort(); d2h.sort()
        return h2d[len(h2d) // 2], d2h[len(d2h) // 2]
