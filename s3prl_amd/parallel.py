"""Data-parallel encoding of one batch over the GPUs of a node: one process per GPU, utterances sharded in
contiguous blocks, hidden states re-assembled with per-layer all-gathers over RCCL (``torch.distributed`` backend
"nccl" on ROCm) — SURVEY §8e.

Reference quirks that couple an utterance to its batch (GroupNorm over the padded length, the ``n_max // T`` frame
mask, T itself) all depend only on the GLOBAL ``n_max``: every rank pads its shard to it, and the
gathered result equals the single-GPU full-batch forward.

The sharding / gathering logic is device-agnostic (``encode_fn`` does the work), so it is covered on CPU with the
gloo backend; on the GPU ``encode_fn`` is ``HipUpstreamExpert.encode`` and the all-gather of layer l is issued on a
side stream as soon as the library signals that hidden_states[l] is final, overlapping the remaining layers.
"""

from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch


def shard_bounds(B: int, world: int, rank: int) -> Tuple[int, int, int]:
    """Contiguous block of utterances for ``rank``: (begin, end, per_rank) with per_rank = ceil(B / world).
    Trailing ranks may own fewer (or zero) real utterances; gathers run on per_rank rows and the pad rows are dropped."""
    per = -(-B // world)
    beg = min(B, rank * per)
    end = min(B, beg + per)
    return beg, end, per


def encode_data_parallel(encode_fn: Callable[[List[torch.Tensor], int], torch.Tensor], wavs: Sequence[torch.Tensor],
                         group=None, overlap_events: Optional[list] = None, algo: str = "ring") -> List[torch.Tensor]:
    """Every rank passes the SAME full list ``wavs`` (only its own shard has to be resident on its device);
    returns ``hidden_states``: NL+1 tensors of shape (B, T, D), identical on every rank, in input order.

    ``encode_fn(shard_wavs, n_max) -> (NL+1, Bs, T, D)`` runs the encoder on the shard padded to ``n_max``.
    """
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = len(wavs)
    n_max = max(int(w.numel()) for w in wavs)
    beg, end, per = shard_bounds(B, world, rank)
    mine = list(wavs[beg:end])
    while len(mine) < per:  # pad the shard with a copy of a real utterance; its rows are dropped after the gather
        mine.append(wavs[min(beg, B - 1)] if not mine else mine[-1])
    hs = encode_fn(mine, n_max)  # (NL+1, per, T, D)
    if world == 1:
        return [hs[l][:B] for l in range(hs.shape[0])]
    gathered = gather_layers(hs, group=group, overlap_events=overlap_events, algo=algo)
    # rank r's block holds utterances [r*per, r*per + per) → already in input order; drop the pad rows at the end
    return [gathered[l][:B] for l in range(gathered.shape[0])]


def featurize_data_parallel(encode_fn: Callable[[List[torch.Tensor], int], torch.Tensor],
                            featurize_fn: Callable[[torch.Tensor], torch.Tensor], wavs: Sequence[torch.Tensor],
                            group=None) -> torch.Tensor:
    """Data-parallel encode + Featurizer with the weighted sum applied BEFORE the exchange: each rank
    reduces its own (NL+1, Bs, T, D) slab to (Bs, T, D) with ``featurize_fn`` and ONE all-gather reassembles the batch —
    (NL+1) x less xGMI traffic than gathering every layer.  Returns (B, T, D), identical on every rank, input order."""
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = len(wavs)
    n_max = max(int(w.numel()) for w in wavs)
    beg, end, per = shard_bounds(B, world, rank)
    mine = list(wavs[beg:end])
    while len(mine) < per:
        mine.append(wavs[min(beg, B - 1)] if not mine else mine[-1])
    feat = featurize_fn(encode_fn(mine, n_max)).contiguous()  # (per, T, D)
    if world == 1:
        return feat[:B]
    out = torch.empty((world * per,) + tuple(feat.shape[1:]), dtype=feat.dtype, device=feat.device)
    dist.all_gather_into_tensor(out, feat, group=group)
    return out[:B]


def featurized_data_parallel(expert, weights: Sequence[float], wavs: Sequence[torch.Tensor], normalize: bool = False,
                             group=None) -> torch.Tensor:
    """Data-parallel encode with the Featurizer's weighted sum computed INSIDE the encoder (``s3enc_forward_ex``
    featurize epilogue): every rank produces only its shard's (Bs, T, D) weighted sum and ONE all-gather
    reassembles the batch — the per-layer slab is neither written nor exchanged.  ``weights``: one float per layer
    (softmax already applied).  Returns (B, T, D) fp32, identical on every rank, input order."""
    return featurize_data_parallel(lambda mine, n_max: expert.encode_featurized(mine, weights, normalize, n_max=n_max),
                                   lambda feat: feat, wavs, group)


_COMM_STREAMS = {}


EXCHANGE_ALGOS = ("ring", "direct")


def _direct_exchange(out_l: torch.Tensor, hs_l: torch.Tensor, group, world: int, rank: int):
    """All-pairs form of one state's exchange: rank r sends its block to every peer and receives theirs with world-1
    point-to-point pairs (peer = rank +- p at step p), batched into ONE group.  xGMI is point-to-point — each pair of GPUs
    has its own link — so this drives all 7 links of a GPU at once where a ring all-gather is bound by one.
    Same bytes in the same places as ``all_gather_into_tensor``; works on every backend (gloo: the CPU tests)."""
    import torch.distributed as dist

    Bs = hs_l.shape[0]
    own = out_l[rank * Bs:(rank + 1) * Bs]
    if own.data_ptr() != hs_l.data_ptr():
        own.copy_(hs_l)
    ops = []
    for pstep in range(1, world):
        to, frm = (rank + pstep) % world, (rank - pstep) % world
        ops.append(dist.P2POp(dist.isend, hs_l, dist.get_global_rank(group, to) if group is not None else to, group))
        ops.append(dist.P2POp(dist.irecv, out_l[frm * Bs:(frm + 1) * Bs],
                              dist.get_global_rank(group, frm) if group is not None else frm, group))
    return dist.batch_isend_irecv(ops) if ops else []


def gather_layers(hs: torch.Tensor, group=None, overlap_events: Optional[list] = None,
                  out: Optional[torch.Tensor] = None, algo: str = "ring") -> torch.Tensor:
    """All-gather a rank-local (NL+1, Bs, T, D) slab into (NL+1, world*Bs, T, D): ONE exchange per layer, so every
    ``hidden_states[l]`` comes out as a contiguous (B, T, D) block in rank order (a single flat gather would be
    rank-major across layers, SURVEY §7 hard part 5).  With ``overlap_events`` (CUDA events recorded by the encoder
    when layer l is final) the exchange of layer l is issued on a side stream and overlaps the remaining compute.
    ``algo``: "ring" = one ``all_gather_into_tensor`` per layer (RCCL's own choice of algorithm), "direct" = the all-pairs
    send / receive form (``_direct_exchange``)."""
    import torch.distributed as dist

    if algo not in EXCHANGE_ALGOS:
        raise ValueError(f"algo must be one of {EXCHANGE_ALGOS}, got {algo!r}")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    NLp1, Bs, T, D = hs.shape
    if out is None:
        out = torch.empty((NLp1, world * Bs, T, D), dtype=hs.dtype, device=hs.device)
    result = out
    if hs.dtype in (torch.bfloat16, torch.float16):
        # 16-bit states (s3enc_forward_ex out_dtype): an all-gather only moves bytes, so exchange them as uint8 — the one
        # element type every backend supports (gloo rejects bfloat16 and int16 on device tensors)
        hs, out = hs.view(torch.uint8), out.view(torch.uint8)

    def one(l):
        if algo == "direct":
            return _direct_exchange(out[l], hs[l], group, world, rank)
        return [dist.all_gather_into_tensor(out[l], hs[l], group=group, async_op=True)]

    works = []
    if overlap_events is not None and hs.is_cuda:
        if len(overlap_events) < NLp1:
            raise ValueError(f"{len(overlap_events)} overlap events for {NLp1} states")
        key = hs.device.index
        if key not in _COMM_STREAMS:
            _COMM_STREAMS[key] = torch.cuda.Stream(device=hs.device)
        comm = _COMM_STREAMS[key]
        for l in range(NLp1):
            with torch.cuda.stream(comm):
                comm.wait_event(overlap_events[l])
                works += one(l)
    else:
        for l in range(NLp1):
            works += one(l)
    for w in works:
        w.wait()
    return result


class RcclComm:
    """The library's own exchange (``s3enc_comm_*`` in include/s3enc.h: RCCL reached from ``libs3enc.so``, one all-gather
    per state on the communicator's stream, each ordered after the encoder's "state l final" event) — the path a non-Python
    binder uses, behind the same call shape as ``gather_layers``.  The 128-byte RCCL id travels over ``torch.distributed``
    when a process group exists (any backend), else ``world`` must be 1."""

    def __init__(self, device: Optional[int] = None, group=None):
        import ctypes as C

        import torch.distributed as dist

        from . import _lib

        self._lib, self._C = _lib.load(), C
        self.device = torch.cuda.current_device() if device is None else int(device)
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        ident = C.create_string_buffer(128)
        if rank == 0:
            _lib.check(self._lib.s3enc_comm_unique_id(ident), "s3enc_comm_unique_id")
        if world > 1:
            box = [bytes(ident.raw)]
            # src is a GLOBAL rank: a subgroup need not contain global rank 0
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = C.create_string_buffer(box[0], 128)
        h = C.c_void_p()
        _lib.check(self._lib.s3enc_comm_init_rank(ident, world, rank, self.device, C.byref(h)), "s3enc_comm_init_rank")
        self._h, self.world, self.rank, self._group = h, world, rank, group
        self._checked_shapes = set()

    def _check_equal_slabs(self, shape):
        """The exchange moves the same byte count from every rank: all ranks must hold the same (NS, Bs, T, D) (a ragged
        last shard is padded to ceil(B / world) by ``encode_data_parallel`` before it gets here).  Checked once per shape."""
        if self.world == 1 or shape in self._checked_shapes:
            return
        import torch.distributed as dist

        shapes = [None] * self.world
        dist.all_gather_object(shapes, tuple(shape), group=self._group)
        if any(tuple(sh) != tuple(shape) for sh in shapes):
            raise ValueError(f"RcclComm.gather_layers: slab shapes differ across ranks: {shapes}")
        self._checked_shapes.add(shape)

    def gather_layers(self, hs: torch.Tensor, overlap_events: Optional[list] = None, out: Optional[torch.Tensor] = None,
                      algo: str = "ring") -> torch.Tensor:
        """(NS, Bs, T, D) on this rank -> (NS, world * Bs, T, D); asynchronous on the current stream.
        ``algo``: "ring" (ncclAllGather) or "direct" (grouped all-pairs ncclSend / ncclRecv) — S3ENC_EXCHANGE_* of the C ABI."""
        C = self._C
        from . import _lib

        if algo not in EXCHANGE_ALGOS:
            raise ValueError(f"algo must be one of {EXCHANGE_ALGOS}, got {algo!r}")
        NS = hs.shape[0]
        per_state = hs[0].numel() * hs.element_size()
        assert hs.is_cuda and hs[0].is_contiguous() and hs.stride(0) * hs.element_size() >= per_state
        self._check_equal_slabs(tuple(hs.shape))
        if out is None:
            out = torch.empty((NS, self.world * hs.shape[1]) + tuple(hs.shape[2:]), dtype=hs.dtype, device=hs.device)
        evs = None
        if overlap_events is not None:
            if len(overlap_events) < NS:
                raise ValueError(f"{len(overlap_events)} overlap events for {NS} states")
            evs = (C.c_void_p * NS)(*[int(ev.cuda_event) for ev in overlap_events[:NS]])
        stream = torch.cuda.current_stream(hs.device).cuda_stream
        _lib.check(self._lib.s3enc_comm_exchange_states(self._h, EXCHANGE_ALGOS.index(algo), C.c_void_p(hs.data_ptr()),
                                                        hs.stride(0) * hs.element_size(), C.c_void_p(out.data_ptr()),
                                                        out.stride(0) * out.element_size(), NS, per_state, evs,
                                                        C.c_void_p(stream)), "s3enc_comm_exchange_states")
        return out

    def close(self):
        if self._h:
            self._lib.s3enc_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DataParallelUpstream(torch.nn.Module):
    """Wraps a ``HipUpstreamExpert``: same ``forward(wavs) -> dict`` contract, batch sharded over the process group."""

    def __init__(self, expert, group=None, overlap: bool = True, algo: str = "ring"):
        super().__init__()
        self.expert = expert
        self.group = group
        self.overlap = overlap
        self.algo = algo

    def get_downsample_rates(self, key: str = None) -> int:
        return self.expert.get_downsample_rates(key)

    def forward(self, wavs):
        events = None
        if self.overlap and wavs[0].is_cuda:
            events = self.expert._encoder_for(wavs[0].device).layer_events()
        hidden = tuple(encode_data_parallel(self.expert.encode, wavs, self.group, events, self.algo))
        result = {"hidden_states": hidden, "last_hidden_state": hidden[-1]}
        for i, h in enumerate(hidden):
            result[f"hidden_state_{i}"] = h
        return result


class CopyComm:
    """S3ENC_EXCHANGE_COPY of the C ABI (include/s3enc.h, csrc/comm.hip): the per-state exchange on the copy engines — every rank's
    block is written into the peers' receive slabs by ``hipMemcpyAsync`` through IPC mappings, one stream per peer (all xGMI links
    at once), each copy behind the encoder's "state l final" event, and no compute unit runs a collective's kernel beside the
    encoder's GEMMs.  No RCCL: the ranks meet through ``torch.distributed`` of ANY backend (the 256-byte handles travel by
    ``all_gather_object``), one PROCESS per GPU.  The receive slab is allocated once per (shape, dtype) and re-used by every
    exchange — ``gather_layers`` returns it, and its contents are valid until the next exchange of that shape."""

    def __init__(self, device: Optional[int] = None, group=None):
        import ctypes as C

        import torch.distributed as dist

        from . import _lib

        self._lib, self._C, self._group = _lib.load(), C, group
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._comms = {}  # (shape, dtype) -> (handle, slab): one registered slab per communicator

    def _comm_for(self, shape, dtype):
        C = self._C
        from . import _lib

        key = (tuple(shape), dtype)
        if key in self._comms:
            return self._comms[key]
        import torch.distributed as dist

        h = C.c_void_p()
        _lib.check(self._lib.s3enc_comm_init_local(self.world, self.rank, self.device, C.byref(h)), "s3enc_comm_init_local")
        NS, Bs = shape[0], shape[1]
        slab = torch.empty((NS, self.world * Bs) + tuple(shape[2:]), dtype=dtype, device=torch.device("cuda", self.device))
        if self.world > 1:
            blob = C.create_string_buffer(_lib.COPY_HANDLE_BYTES)
            _lib.check(self._lib.s3enc_comm_copy_export(h, C.c_void_p(slab.data_ptr()), slab.numel() * slab.element_size(), blob),
                       "s3enc_comm_copy_export")
            blobs = [None] * self.world
            dist.all_gather_object(blobs, (tuple(shape), str(dtype), bytes(blob.raw)), group=self._group)
            if any(b[:2] != (tuple(shape), str(dtype)) for b in blobs):
                raise ValueError(f"CopyComm: slab shapes differ across ranks: {[b[:2] for b in blobs]}")
            allb = C.create_string_buffer(b"".join(b[2] for b in blobs), _lib.COPY_HANDLE_BYTES * self.world)
            _lib.check(self._lib.s3enc_comm_copy_attach(h, allb), "s3enc_comm_copy_attach")
        self._comms[key] = (h, slab)
        return h, slab

    def gather_layers(self, hs: torch.Tensor, overlap_events: Optional[list] = None) -> torch.Tensor:
        """(NS, Bs, T, D) on this rank -> the registered (NS, world * Bs, T, D) slab; asynchronous on the current stream."""
        C = self._C
        from . import _lib

        NS = hs.shape[0]
        per_state = hs[0].numel() * hs.element_size()
        assert hs.is_cuda and hs[0].is_contiguous() and hs.stride(0) * hs.element_size() >= per_state
        h, slab = self._comm_for(hs.shape, hs.dtype)
        evs = None
        if overlap_events is not None:
            if len(overlap_events) < NS:
                raise ValueError(f"{len(overlap_events)} overlap events for {NS} states")
            evs = (C.c_void_p * NS)(*[int(ev.cuda_event) for ev in overlap_events[:NS]])
        stream = torch.cuda.current_stream(hs.device).cuda_stream
        _lib.check(self._lib.s3enc_comm_exchange_states(h, _lib.EXCHANGE_COPY, C.c_void_p(hs.data_ptr()), hs.stride(0) * hs.element_size(),
                                                        C.c_void_p(slab.data_ptr()), slab.stride(0) * slab.element_size(), NS, per_state,
                                                        evs, C.c_void_p(stream)), "s3enc_comm_exchange_states")
        return slab

    def release(self) -> None:
        """Call BEFORE enqueueing a forward whose states the next ``gather_layers`` moves: "whoever reads the slabs' current
        contents has been enqueued on the current stream" — the peers may then overwrite them while that forward still computes
        (the per-state pushes overlap it).  Without it an exchange counts from its own call: correct, not overlapped."""
        stream = torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream
        from . import _lib

        for h, _ in self._comms.values():
            _lib.check(self._lib.s3enc_comm_copy_release(h, self._C.c_void_p(stream)), "s3enc_comm_copy_release")

    def status(self) -> int:
        """0, or a bit mask of the peers a wait gave up on (deadline S3ENC_COPY_DEADLINE_MS); synchronises the exchange streams."""
        C = self._C
        from . import _lib

        out = 0
        for h, _ in self._comms.values():
            st = C.c_int32()
            _lib.check(self._lib.s3enc_comm_copy_status(h, C.byref(st)), "s3enc_comm_copy_status")
            out |= st.value
        return out

    def close(self):
        for h, _ in self._comms.values():
            self._lib.s3enc_comm_destroy(h)
        self._comms = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
