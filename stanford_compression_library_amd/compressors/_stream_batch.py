"""Batched form of the block loop ``DataEncoder.encode`` / ``DataDecoder.decode`` -- row f2 of the scope table.

The reference walks a stream block by block: ``get_block`` -> ``encode_block`` -> ``write_block`` (one framed
record per block, scl/core/data_encoder_decoder.py:57-69 and :118-144).  The static-model coders make every block
independent, so here the whole stream becomes ONE batched launch (one lane per block) followed by the device-side
framing pass (``scl_streams_compact`` with ``SCL_COMPACT_FRAMED``), and the file bytes are written in one go.
The resulting file is byte-identical to the per-block loop (and to the reference's ``EncodedBlockWriter``).

Only ``write_block`` / ``get_block`` are part of the reference's writer / reader contract: a writer without
``write_framed_bytes`` or a reader without ``file_reader`` gets the inherited per-block loop.  Streams are processed in
bounded batches (``MAX_BATCH_BYTES`` of symbols at a time).  On the decode side the file is untrusted input: every length
is checked with a real ``raise`` (``python -O`` strips ``assert`` statements), a launch never allocates more than
``MAX_BATCH_BYTES`` of output for its blocks together (n_blocks x largest header, not just per block), and a block whose
header announces more than ``MAX_BLOCK_SYMBOLS`` -- valid for the reference, whose DATA_BLOCK_SIZE_BITS = 32 allows it --
is decoded on its own through ``decode_block`` instead of being refused.

Two host shapes feed those launches:

* the LIST shape -- any stream with ``get_block`` / ``write_block``: ``DataBlock`` lists in, ``DataBlock`` lists out (the
  reference's contract; a Python object per symbol, ~35 MB/s end to end);
* the BULK shape (round 5) -- streams that also move symbols as arrays of integer codes (``read_codes`` / ``write_codes``:
  ``Uint8FileDataStream``, ``TextFileDataStream``; core/data_stream.py): slabs of ``SLAB_BYTES`` symbols are read straight
  into pinned staging buffers, mapped to alphabet indices on the DEVICE (a 256-entry table built from the model's own
  symbol -> index dict, so an unknown symbol raises the same ``KeyError``), encoded, framed, and copied back while the host
  reads the next slab; on the way back the record index of a whole slab comes from ONE call of the C ABI
  (``scl_framed_index_host``: the walk ``EncodedBlockReader.get_block`` makes record by record), the decoded rows are
  packed and mapped back to codes on the device and written with one ``write``.  Same file bytes, same exceptions; no
  Python object per symbol (tools/time_stream_file.py).
"""
from __future__ import annotations

import numpy as np

from ..backend.models import compact, compact_capacity, compact_into, compact_scratch_bytes, framed_index_host
from ..utils import fileio as _fileio

def _stock_read_codes(data_stream) -> bool:
    from ..core.data_stream import Uint8FileDataStream

    return getattr(type(data_stream), "read_codes", None) is Uint8FileDataStream.read_codes


MAX_BATCH_BYTES = 1 << 28     # symbols per launch: bounds host + device memory whatever the stream length
MAX_BLOCK_SYMBOLS = 1 << 26   # blocks announcing more symbols than this are decoded one by one (decode_block)
SLAB_BYTES = 1 << 26          # bulk shape: symbols (encode) / file bytes (decode) per pipeline stage
PIN_MIN_BYTES = 1 << 20       # staging buffers below this are ordinary host memory (pinning costs more than it saves)


def _check(cond, msg):
    """AssertionError like the reference's own checks (data_encoder_decoder.py:141), but not an ``assert`` statement"""
    if not cond:
        raise AssertionError(msg)


def _bulk_capable(stream, bulk: str, per_symbol) -> bool:
    """does ``stream`` offer the bulk method AND still move symbols the way the class that defines it does?  A subclass that
    overrides ``get_block`` / ``get_symbol`` (``write_block`` / ``write_symbol``) -- a filter, a counter, a transform --
    must see every symbol: it gets the LIST shape."""
    cls = type(stream)
    owner = next((k for k in cls.__mro__ if bulk in k.__dict__), None)
    return owner is not None and all(getattr(cls, m, None) is getattr(owner, m, None) for m in per_symbol)


def _host_buffer(nbytes: int):
    """-> (torch uint8 tensor, numpy view of it); page-locked when it is big enough to be worth it"""
    import torch

    t = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, pin_memory=nbytes >= PIN_MIN_BYTES)
    return t, t.numpy()


def _read_full(read_piece, n: int):
    """call ``read_piece(k)`` (-> array of at most k codes, or None at the end) until n codes are there or the stream ends:
    a short read in the middle of a stream (pipes, text decoders) must not move a block boundary"""
    first = read_piece(n)
    if first is None or len(first) == n:
        return first
    pieces, have = [first], len(first)
    while have < n:
        nxt = read_piece(n - have)
        if nxt is None:
            break
        pieces.append(nxt)
        have += len(nxt)
    if len(pieces) == 1:
        return first
    dt = np.uint32 if any(p.dtype != np.uint8 for p in pieces) else np.uint8
    return np.concatenate([p.astype(dt, copy=False) for p in pieces])


def _fill(fobj, h: np.ndarray, n: int) -> int:
    """``readinto`` until h[:n] is full or the file ends -> bytes read (regular files: several positional reads at once,
    utils/fileio.py)"""
    view = memoryview(h)
    got = _fileio.read_into(fobj, view, n)
    if got is not None:
        return got
    got = 0
    while got < n:
        k = fobj.readinto(view[got:n])
        if not k:
            break
        got += k
    return got


def _fill_from(fobj, h: np.ndarray, start: int) -> int:
    """``readinto`` behind h[:start] until the buffer is full or the file ends -> bytes read"""
    view = memoryview(h)
    par = _fileio.read_into(fobj, view[start:], h.size - start)
    if par is not None:
        return par
    got = start
    while got < h.size:
        k = fobj.readinto(view[got:])
        if not k:
            break
        got += k
    return got - start


def _bytes_left(fobj):
    """bytes between the file position and the end of a regular file, or None when that cannot be known"""
    import os
    import stat

    try:
        st = os.fstat(fobj.fileno())
        if not stat.S_ISREG(st.st_mode):
            return None
        return max(0, st.st_size - fobj.tell())
    except (OSError, AttributeError, ValueError):
        return None


class _EncodeStage:
    """device + staging buffers of one pipeline stage of the bulk encoder (two stages alternate)"""

    def __init__(self, model, n_blocks: int, block_size: int, dev):
        import torch

        self.n_blocks, self.block_size = n_blocks, block_size
        self.h_in_t, self.h_in = _host_buffer(n_blocks * block_size)
        self.d_raw = torch.empty(n_blocks * block_size, dtype=torch.uint8, device=dev)
        self.enc = model.alloc_encoded(n_blocks, block_size, dev, model.slot_bytes(block_size))
        cap = compact_capacity(n_blocks, self.enc.stride, framed=True)
        self.d_framed = torch.empty(cap, dtype=torch.uint8, device=dev)
        self.d_offs = torch.empty(n_blocks + 1, dtype=torch.int64, device=dev)
        self.d_scratch = torch.empty(compact_scratch_bytes(n_blocks), dtype=torch.uint8, device=dev)
        self.h_out_t = self.h_out = None  # sized by the first result (the worst case is 1.5x the typical one)
        self.busy = None                  # (event after the launch, n_blocks, EncodedBatch, offsets view)
        self.h2d_done = None              # event after the copy out of h_in: the reader thread may refill it
        self.written = None               # future of the file write out of h_out

    def host_out(self, nbytes: int):
        if self.written is not None:  # the writer thread still reads h_out
            self.written.result()
            self.written = None
        if self.h_out is None or self.h_out.size < nbytes:
            self.h_out_t, self.h_out = _host_buffer(nbytes + nbytes // 8 + 64)
        return self.h_out_t, self.h_out


class BatchedStreamEncoderMixin:
    """expects ``self._batch_model()`` -> (device model, symbol->index dict)"""

    def encode(self, data_stream, block_size: int, encode_writer):
        if not hasattr(encode_writer, "write_framed_bytes"):
            return super().encode(data_stream, block_size, encode_writer)  # any writer with write_block
        if hasattr(data_stream, "symbol_of") and _bulk_capable(data_stream, "read_codes", ("get_block", "get_symbol")):
            return self._encode_bulk(data_stream, block_size, encode_writer)
        per_batch = max(1, MAX_BATCH_BYTES // max(1, block_size))
        while True:
            blocks = []
            while len(blocks) < per_batch:
                blk = data_stream.get_block(block_size)
                if blk is None:
                    break
                blocks.append(blk)
            if blocks:
                self._encode_batch(blocks, block_size, encode_writer)
            if len(blocks) < per_batch:
                return

    def _encode_batch(self, blocks, block_size: int, encode_writer):
        import torch

        model, index_of = self._batch_model()
        n = len(blocks)
        width = (block_size + 15) // 16 * 16
        sym = np.zeros((n, width), dtype=model.sym_dtype)  # uint8, or uint16 for alphabets above 256 symbols
        lens = np.zeros(n, dtype=np.int32)
        for i, blk in enumerate(blocks):
            data = blk.data_list
            try:
                sym[i, :len(data)] = np.fromiter((index_of[s] for s in data), dtype=model.sym_dtype, count=len(data))
            except KeyError:
                # the reference's loop has written blocks 0 .. i-1 when encode_block(block i) raises: so do we
                if i:
                    self._encode_batch(blocks[:i], block_size, encode_writer)
                raise
            lens[i] = len(data)
        dev = torch.device("cuda", torch.cuda.current_device())
        enc = model.encode_batch(torch.from_numpy(sym).to(dev)[:, :block_size] if width == block_size
                                 else torch.from_numpy(sym).to(dev), lens=torch.from_numpy(lens).to(dev),
                                 out_stride=model.slot_bytes(block_size))
        framed, offs = compact(enc, framed=True)
        status = enc.status.cpu().numpy()
        _check(not status.any(), f"device reported chunk status {status[status != 0][:4]}")
        encode_writer.write_framed_bytes(framed[: int(offs[-1].item())].cpu().numpy().tobytes())

    # -- bulk shape -------------------------------------------------------------------------------------------------
    def _encode_bulk(self, data_stream, block_size: int, encode_writer):
        """Three actors: a reader thread fills the staging buffer of one stage (file reads release the GIL), this thread
        drives the device for the other, a writer thread writes the bytes of the stage before.  The device work of a slab
        (copy in, table lookup, encode, framing, copy out: a few milliseconds per 64 MiB) hides under the file I/O."""
        from concurrent.futures import ThreadPoolExecutor

        _check(block_size >= 1, "block_size must be positive")
        rd, wr = ThreadPoolExecutor(1), ThreadPoolExecutor(1)
        try:
            self._encode_bulk_loop(data_stream, block_size, encode_writer, rd, wr)
        finally:
            rd.shutdown(wait=True)
            wr.shutdown(wait=True)

    def _encode_bulk_loop(self, data_stream, block_size: int, encode_writer, rd, wr):
        import torch

        model, index_of = self._batch_model()
        dev = torch.device("cuda", torch.cuda.current_device())
        per_slab = max(1, min(SLAB_BYTES, MAX_BATCH_BYTES) // block_size)
        n_sym = per_slab * block_size
        fobj = getattr(data_stream, "file_obj", None)
        # straight-into-the-staging-buffer reads only for a plain binary file behind the stock Uint8FileDataStream: a file
        # object whose ``mode`` is not a string (gzip.GzipFile: 1 / 2) or a subclass with its own ``read_codes`` goes through
        # ``read_codes`` like any other stream
        mode = getattr(fobj, "mode", "")
        byte_file = (fobj is not None and isinstance(mode, str) and "b" in mode and hasattr(fobj, "readinto")
                     and _stock_read_codes(data_stream))
        sized = byte_file and _bytes_left(fobj) is not None  # small files: small staging buffers

        def new_stage_for_rest():
            want = min(per_slab, max(1, ((_bytes_left(fobj) or 0) + block_size - 1) // block_size))
            return _EncodeStage(model, want, block_size, dev)

        def read_slab(stage):  # reader thread -> (codes | None, whether they already sit in stage.h_in)
            if byte_file and stage is not None:  # bytes go straight into the (pinned) staging buffer
                room = min(n_sym, stage.n_blocks * block_size)
                n = _fill(fobj, stage.h_in, room)
                if n == room and room < n_sym:  # the size hint was short (the file grew): finish the slab the slow way
                    more = _read_full(data_stream.read_codes, n_sym - n)
                    if more is not None:
                        return np.concatenate([stage.h_in[:n], more]), False
                return (stage.h_in[:n] if n else None), True
            return _read_full(data_stream.read_codes, n_sym), False

        stages, lut8, turn, pending, fail = [None, None], None, 0, None, None
        if sized:
            stages[0] = new_stage_for_rest()
        fut = rd.submit(read_slab, stages[0])
        while True:
            codes, in_place = fut.result()
            if codes is None:
                break
            n = int(len(codes))
            last = n < n_sym
            nb = (n + block_size - 1) // block_size
            stage = stages[turn]  # free: its previous results were handed to the writer one iteration ago
            if stage is None or stage.n_blocks < nb:
                stage = stages[turn] = _EncodeStage(model, nb if last else per_slab, block_size, dev)
                in_place = False
            if not last:  # the next slab is read while this one is on the device
                other = stages[turn ^ 1]
                if other is None and sized:
                    other = stages[turn ^ 1] = new_stage_for_rest()
                if other is not None and other.h2d_done is not None:
                    other.h2d_done.synchronize()  # (long over: its launch was a whole slab ago)
                fut = rd.submit(read_slab, other)
            # ---- symbols -> alphabet indices on the device ----------------------------------------------------
            # a symbol the model does not know: the reference's block loop has written every block in front of the
            # offending one when its encode_block raises KeyError -- so the whole blocks in front of it in this slab are
            # still encoded and written (with the pending slab) before the KeyError leaves
            fail = None
            if codes.dtype == np.uint8:
                if lut8 is None:
                    lut8 = self._code_lut8(data_stream, index_of, model, dev)
                if not in_place:
                    stage.h_in[:n] = codes
                d_raw = stage.d_raw[: nb * block_size]
                d_raw[:n].copy_(stage.h_in_t[:n], non_blocking=True)
                stage.h2d_done = torch.cuda.Event()
                stage.h2d_done.record()
                if n < nb * block_size:
                    d_raw[n:].zero_()
                lut_dev, identity = lut8
                if identity:
                    d_idx = d_raw
                else:
                    wide_idx = lut_dev[d_raw[:n].to(torch.int64)]
                    bad = (wide_idx < 0).nonzero()
                    if bad.numel():  # the first symbol the model does not know, as the per-block loop would meet it
                        first = int(bad[0].item())
                        fail = KeyError(data_stream.symbol_of(int(stage.h_in[first])))
                        nb = first // block_size
                        n = nb * block_size
                        wide_idx = wide_idx[:n]
                    d_idx = torch.zeros(max(nb, 1) * block_size, dtype=self._torch_sym_dtype(model), device=dev)
                    d_idx[:n] = wide_idx.to(d_idx.dtype)
            else:  # code points beyond a byte (rare): mapped on the host through the codes that occur
                uniq, inv = np.unique(codes, return_inverse=True)
                vals = np.array([index_of.get(data_stream.symbol_of(int(u)), -1) for u in uniq], dtype=np.int64)
                mapped = vals[inv]
                if (vals < 0).any():
                    first = int(np.flatnonzero(mapped < 0)[0])
                    fail = KeyError(data_stream.symbol_of(int(codes[first])))
                    nb = first // block_size
                    n = nb * block_size
                    mapped = mapped[:n]
                h_idx = np.zeros(max(nb, 1) * block_size, dtype=model.sym_dtype)
                h_idx[:n] = mapped
                d_idx = torch.from_numpy(h_idx.view(np.int16) if h_idx.dtype == np.uint16 else h_idx).to(dev)
            if nb:
                lens = torch.full((nb,), block_size, dtype=torch.int32, device=dev)
                if n < nb * block_size:
                    lens[-1] = n - (nb - 1) * block_size
                enc = stage.enc if nb == stage.n_blocks else model.alloc_encoded(nb, block_size, dev,
                                                                                 model.slot_bytes(block_size))
                model.encode_batch(d_idx[: nb * block_size].view(nb, block_size), lens=lens, out=enc)
                d_offs = stage.d_offs[: nb + 1]
                compact_into(enc, stage.d_framed, d_offs, stage.d_scratch, framed=True)
                ev = torch.cuda.Event()
                ev.record()
                stage.busy = (ev, nb, enc, d_offs)
            if pending is not None:  # the previous slab's bytes leave while this one is on the device
                self._finish_encode_stage(pending, encode_writer, wr)
            pending = stage if nb else None
            turn ^= 1
            if last or fail is not None:
                break
        if pending is not None:
            self._finish_encode_stage(pending, encode_writer, wr)
        for st in stages:
            if st is not None and st.written is not None:
                st.written.result()  # (re-raises what the writer thread met)
                st.written = None
        if fail is not None:
            raise fail

    @staticmethod
    def _torch_sym_dtype(model):
        import torch

        return torch.uint8 if model.sym_dtype == np.uint8 else torch.int16  # (torch has no uint16 arithmetic: same bits)

    @staticmethod
    def _code_lut8(data_stream, index_of, model, dev):
        """code (0..255) -> alphabet index through the model's own dict (what ``index_of[symbol]`` gives the per-block loop);
        -1 = a symbol the model does not know.  -> (int32 device table, whether it is the identity)"""
        import torch

        lut = np.array([index_of.get(data_stream.symbol_of(c), -1) for c in range(256)], dtype=np.int32)
        identity = model.sym_dtype == np.uint8 and bool((lut == np.arange(256)).all())
        return torch.from_numpy(lut).to(dev), identity

    @staticmethod
    def _finish_encode_stage(stage, encode_writer, wr):
        ev, nb, enc, d_offs = stage.busy
        ev.synchronize()
        status = enc.status[:nb]
        if bool(status.any().item()):
            bad = status[status != 0][:4].cpu().numpy()
            raise AssertionError(f"device reported chunk status {bad}")
        total = int(d_offs[nb].item())
        h_t, h = stage.host_out(total)
        h_t[:total].copy_(stage.d_framed[:total])
        stage.written = wr.submit(encode_writer.write_framed_bytes, memoryview(h[:total]))
        stage.busy = None


class _CodeSink:
    """ordered writes of decoded codes: packed runs go through two page-locked buffers to a writer thread"""

    def __init__(self, output_stream, wr):
        self.out, self.wr = output_stream, wr
        self.bufs, self.futs, self.turn = [None, None], [None, None], 0

    def drain(self):
        for i in (0, 1):
            if self.futs[i] is not None:
                self.futs[i].result()
                self.futs[i] = None

    def write_device_bytes(self, d_bytes):
        k, n = self.turn, int(d_bytes.numel())
        if self.futs[k] is not None:
            self.futs[k].result()
            self.futs[k] = None
        if self.bufs[k] is None or self.bufs[k][1].size < n:
            self.bufs[k] = _host_buffer(n + n // 8)
        h_t, h = self.bufs[k]
        h_t[:n].copy_(d_bytes)  # through a page-locked buffer: three times the rate of a pageable copy
        self.futs[k] = self.wr.submit(self.out.write_codes, h[:n])
        self.turn ^= 1

    def write_codes(self, codes: np.ndarray):
        self.drain()
        self.out.write_codes(codes)

    def write_block(self, block):
        self.drain()
        self.out.write_block(block)


class BatchedStreamDecoderMixin:
    """expects ``self._batch_model()`` -> (device model, alphabet list) and ``self._size_bits``"""

    def decode(self, encode_reader, output_stream):
        if not hasattr(encode_reader, "file_reader"):
            return super().decode(encode_reader, output_stream)  # any reader with get_block
        from concurrent.futures import ThreadPoolExecutor

        rd, wr = ThreadPoolExecutor(1), ThreadPoolExecutor(1)
        try:
            sink = _CodeSink(output_stream, wr)
            self._decode_loop(encode_reader.file_reader, sink, rd)
            sink.drain()
        finally:
            rd.shutdown(wait=True)
            wr.shutdown(wait=True)

    def _decode_loop(self, f, sink, rd):
        """slabs of the file through two staging buffers: the reader thread fills one (behind the record that crossed the
        end of the other) while this thread indexes, uploads and decodes the other"""
        self._batch_model()  # creates the device model and, with it, self._size_bits
        sb = self._size_bits
        left = _bytes_left(f)
        cap = max(int(SLAB_BYTES) if left is None else min(int(SLAB_BYTES), left), 64)

        def fill(h, start):  # reader thread -> (bytes in the buffer, whether the file ended before it was full)
            got = _fill_from(f, h, start)
            return start + got, start + got < h.size

        bufs = [_host_buffer(cap), None]
        turn = 0
        fut = rd.submit(fill, bufs[0][1], 0)
        while True:
            filled, eof = fut.result()
            if filled == 0:
                return
            h_t, h = bufs[turn]
            offs, nbits, sizes, used, bad = framed_index_host(h[:filled], sb, partial=True)
            if bad is not None:
                # a malformed record: the reference's block loop has decoded and written every record in front of it by the
                # time it raises (data_encoder_decoder.py:118-144) -- so do that first
                if len(offs):
                    self._decode_slab(h_t, h, used, offs, nbits, sizes, sink)
                sink.drain()
                raise AssertionError(bad)
            if len(offs) == 0:
                _check(not eof, "truncated block file")
                # one record larger than the staging buffer: make room for it (header says how much) and read on
                payload = int.from_bytes(h[:4].tobytes(), "big") if filled >= 4 else 0
                rest = _bytes_left(f)
                # (untrusted header: never allocate for bytes the file does not have)
                _check(rest is None or 4 + payload - filled <= rest, "truncated block file")
                big = _host_buffer(2 * h.size if rest is None else max(h.size + 64, 4 + payload + 64))
                big[1][:filled] = h[:filled]
                bufs[turn] = big
                fut = rd.submit(fill, big[1], filled)
                continue
            carry = filled - used
            _check(not (eof and carry), "truncated block file")
            if not eof:  # the record that crosses the end of this buffer opens the next one
                other = turn ^ 1
                if bufs[other] is None or bufs[other][1].size < max(h.size, cap):
                    bufs[other] = _host_buffer(max(h.size, cap))
                bufs[other][1][:carry] = h[used:filled]
                fut = rd.submit(fill, bufs[other][1], carry)
            self._decode_slab(h_t, h, used, offs, nbits, sizes, sink)
            if eof:
                return
            turn ^= 1

    def _decode_slab(self, h_t, h, used: int, offs, nbits, sizes, sink):
        """decode the records of h[:used] (synchronous: on return the staging buffer is free again)"""
        import torch

        dev = torch.device("cuda", torch.cuda.current_device())
        d_buf = torch.empty(used + 64, dtype=torch.uint8, device=dev)
        d_buf[:used].copy_(h_t[:used], non_blocking=True)
        d_buf[used:].zero_()
        n = len(offs)
        big = sizes > MAX_BLOCK_SYMBOLS
        # launches of at most MAX_BATCH_BYTES of decoded symbols (n_blocks x the largest header among them); a block that
        # announces more than MAX_BLOCK_SYMBOLS goes through decode_block alone
        if not big.any() and n * max(int(sizes.max()), 1) <= MAX_BATCH_BYTES:
            self._decode_group(d_buf, h, offs, nbits, max(int(sizes.max()), 1), sink)
            return
        group = []

        def flush():
            if group:
                g = np.array(group)
                self._decode_group(d_buf, h, offs[g], nbits[g], max(int(sizes[g].max()), 1), sink)
                group.clear()

        gcap = 1
        for i in range(n):
            if big[i]:
                flush()
                gcap = 1
                self._decode_single(h, int(offs[i]), int(nbits[i]), sink)
                continue
            c = max(gcap, int(sizes[i]), 1)
            if group and (len(group) + 1) * c > MAX_BATCH_BYTES:
                flush()
                c = max(int(sizes[i]), 1)
            group.append(i)
            gcap = c
        flush()
        torch.cuda.current_stream(dev).synchronize()  # (the upload, when every record took the one-block path)

    def _decode_single(self, h, off: int, nb: int, sink):
        from ..utils.bitarray_utils import BitArray

        first = off // 8
        ba = BitArray()
        ba.frombytes(h[first:first + (off % 8 + nb + 7) // 8].tobytes())
        block, used = self.decode_block(ba[off % 8:off % 8 + nb])
        _check(used == nb, "num_bits_consumed != len(encoded_block)")  # data_encoder_decoder.py:141
        sink.write_block(block)

    def _decode_group(self, d_buf, h, offs, nbits, cap: int, sink):
        import torch

        from ..core.data_block import DataBlock

        model, alphabet = self._batch_model()
        dev = d_buf.device
        n = len(offs)
        if (nbits >> np.uint64(31)).any():  # the batch entry points carry 32-bit stream lengths
            for i in range(n):
                self._decode_single(h, int(offs[i]), int(nbits[i]), sink)
            return
        d_nbits = torch.from_numpy(nbits.astype(np.int32)).to(dev)
        sym, lens, used, status = model.decode_batch(d_buf, torch.from_numpy(offs.astype(np.int64)).to(dev), d_nbits,
                                                     cap)
        if bool(status.any().item()):
            bad = status[status != 0][:4].cpu().numpy()
            raise AssertionError(f"device reported chunk status {bad}")
        _check(bool((used == d_nbits).all().item()), "num_bits_consumed != len(encoded_block)")  # data_encoder_decoder.py:141
        codes = self._alphabet_codes(sink.out, alphabet, dev)
        if sym.dtype == torch.uint16:  # (torch has next to no uint16 arithmetic: same bits, widened)
            sym = sym.view(torch.int16).to(torch.int32) & 0xFFFF
        if codes is None:  # LIST shape: the stream wants DataBlocks
            lens, sym = lens.cpu().numpy(), sym.cpu().numpy()
            for i in range(n):
                sink.write_block(DataBlock([alphabet[j] for j in sym[i, :lens[i]].tolist()]))
            return
        # BULK shape: rows -> one run of symbols -> codes -> one write
        full = bool((lens[:-1] == cap).all().item()) if n > 1 else True
        if full:
            total = (n - 1) * cap + int(lens[-1].item())
            flat = (sym if sym.is_contiguous() else sym.contiguous()).reshape(-1)[:total]
        else:  # blocks of different sizes in one file
            keep = torch.arange(cap, device=dev)[None, :] < lens[:, None]
            flat = sym[keep]
        lut, identity, np_dtype = codes
        out = flat if identity else lut[flat.to(torch.int64)]
        if out.dtype == torch.uint8:
            sink.write_device_bytes(out)
        else:
            sink.write_codes(out.cpu().numpy().astype(np_dtype))

    def _alphabet_codes(self, output_stream, alphabet, dev):
        """alphabet index -> code table of a stream with ``write_codes`` (None: the stream takes DataBlocks, or some symbol
        is not one it can write in bulk) -> (device table, whether it is the identity, numpy dtype of the codes)"""
        import torch

        if not (hasattr(output_stream, "code_of") and
                _bulk_capable(output_stream, "write_codes", ("write_block", "write_symbol"))):
            return None
        key = (type(output_stream), id(alphabet))
        cache = self.__dict__.setdefault("_code_tables", {})
        if key not in cache:
            raw = [output_stream.code_of(s) for s in alphabet]
            if any(c is None for c in raw):
                cache[key] = None
            else:
                a = np.array(raw, dtype=np.int64)
                np_dtype = np.uint8 if a.max(initial=0) < 256 else np.uint32
                identity = np_dtype == np.uint8 and len(a) <= 256 and bool((a == np.arange(len(a))).all())
                t = torch.from_numpy(a.astype(np.uint8) if np_dtype == np.uint8 else a.astype(np.int32)).to(dev)
                cache[key] = (t, identity, np_dtype)
        return cache[key]
