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
"""
from __future__ import annotations

import numpy as np

from ..backend.models import compact

MAX_BATCH_BYTES = 1 << 28     # symbols per launch: bounds host + device memory whatever the stream length
MAX_BLOCK_SYMBOLS = 1 << 26   # blocks announcing more symbols than this are decoded one by one (decode_block)


def _check(cond, msg):
    """AssertionError like the reference's own checks (data_encoder_decoder.py:141), but not an ``assert`` statement"""
    if not cond:
        raise AssertionError(msg)


class BatchedStreamEncoderMixin:
    """expects ``self._batch_model()`` -> (device model, symbol->index dict)"""

    def encode(self, data_stream, block_size: int, encode_writer):
        if not hasattr(encode_writer, "write_framed_bytes"):
            return super().encode(data_stream, block_size, encode_writer)  # any writer with write_block
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
            sym[i, :len(data)] = np.fromiter((index_of[s] for s in data), dtype=model.sym_dtype, count=len(data))
            lens[i] = len(data)
        dev = torch.device("cuda", torch.cuda.current_device())
        enc = model.encode_batch(torch.from_numpy(sym).to(dev)[:, :block_size] if width == block_size
                                 else torch.from_numpy(sym).to(dev), lens=torch.from_numpy(lens).to(dev),
                                 out_stride=model.slot_bytes(block_size))
        framed, offs = compact(enc, framed=True)
        status = enc.status.cpu().numpy()
        _check(not status.any(), f"device reported chunk status {status[status != 0][:4]}")
        encode_writer.write_framed_bytes(framed[: int(offs[-1].item())].cpu().numpy().tobytes())


class BatchedStreamDecoderMixin:
    """expects ``self._batch_model()`` -> (device model, alphabet list) and ``self._size_bits``"""

    def decode(self, encode_reader, output_stream):
        if not hasattr(encode_reader, "file_reader"):
            return super().decode(encode_reader, output_stream)  # any reader with get_block
        f = encode_reader.file_reader
        while True:
            # one batch = whole framed records up to MAX_BATCH_BYTES (at least one record)
            chunks, size = [], 0
            while size < MAX_BATCH_BYTES:
                header = f.read(4)
                if len(header) == 0:
                    break
                _check(len(header) == 4, "truncated block file")
                payload = int.from_bytes(header, "big")
                body = f.read(payload)
                _check(len(body) == payload and payload >= 1, "truncated block file")
                chunks += [header, body]
                size += 4 + payload
            if not chunks:
                return
            self._decode_batch(np.frombuffer(b"".join(chunks), dtype=np.uint8), output_stream)

    def _decode_batch(self, raw, output_stream):
        from ..core.data_block import DataBlock
        from ..utils.bitarray_utils import BitArray

        self._batch_model()  # creates the device model and, with it, self._size_bits
        # walk the 4-byte block headers on the host (one per block); the payload bits stay where they are
        sb = self._size_bits
        offs, nbits, sizes, pos = [], [], [], 0
        while pos < raw.size:
            _check(pos + 5 <= raw.size, "truncated block file")
            payload = int.from_bytes(raw[pos:pos + 4].tobytes(), "big")
            _check(payload >= 1 and pos + 4 + payload <= raw.size, "truncated block file")
            pad = int(raw[pos + 4]) >> 5
            nb = 8 * payload - 3 - pad
            _check(nb >= sb, "corrupt block: shorter than its DATA_BLOCK_SIZE_BITS header")
            o = 8 * (pos + 4) + 3 + pad
            bits = np.unpackbits(raw[o // 8:(o + sb + 7) // 8 + 1])[o % 8:o % 8 + sb]
            sizes.append(int("".join(map(str, bits.tolist())), 2))  # the block's own size header
            offs.append(o)
            nbits.append(nb)
            pos += 4 + payload
        # launches of at most MAX_BATCH_BYTES of decoded symbols (n_blocks x the largest header among them); a block that
        # announces more than MAX_BLOCK_SYMBOLS goes through decode_block alone
        group = []

        def flush():
            if group:
                self._decode_group(raw, [offs[i] for i in group], [nbits[i] for i in group],
                                   max(sizes[i] for i in group), output_stream)
                group.clear()

        for i, n in enumerate(sizes):
            if n > MAX_BLOCK_SYMBOLS:
                flush()
                first = offs[i] // 8
                ba = BitArray()
                ba.frombytes(raw[first:first + (offs[i] % 8 + nbits[i] + 7) // 8].tobytes())
                block, used = self.decode_block(ba[offs[i] % 8:offs[i] % 8 + nbits[i]])
                _check(used == nbits[i], "num_bits_consumed != len(encoded_block)")  # data_encoder_decoder.py:141
                output_stream.write_block(block)
                continue
            cap = max([sizes[j] for j in group] + [n, 1])
            if group and (len(group) + 1) * cap > MAX_BATCH_BYTES:
                flush()
            group.append(i)
        flush()

    def _decode_group(self, raw, offs, nbits, cap, output_stream):
        import torch

        from ..core.data_block import DataBlock

        model, alphabet = self._batch_model()
        dev = torch.device("cuda", torch.cuda.current_device())
        buf = torch.zeros(raw.size + 64, dtype=torch.uint8, device=dev)
        buf[: raw.size] = torch.from_numpy(raw.copy()).to(dev)
        sym, lens, used, status = model.decode_batch(buf, torch.tensor(offs, dtype=torch.int64, device=dev),
                                                     torch.tensor(nbits, dtype=torch.int32, device=dev), max(cap, 1))
        torch.cuda.synchronize()
        status = status.cpu().numpy()
        _check(not status.any(), f"device reported chunk status {status[status != 0][:4]}")
        used, lens, sym = used.cpu().numpy(), lens.cpu().numpy(), sym.cpu().numpy()
        for i in range(len(offs)):
            _check(int(used[i]) == nbits[i], "num_bits_consumed != len(encoded_block)")  # data_encoder_decoder.py:141
            output_stream.write_block(DataBlock([alphabet[j] for j in sym[i, :lens[i]].tolist()]))
