"""CPU restatement of the reference's chunk codecs -- TEST INFRASTRUCTURE ONLY.

  rle_encode / raw_encode   <- DataChunkSerializer.cs:20-100 (Raw code 0x00, RLE code 0x01; RLE payload =
                               repeated (u32 runLength little-endian, u8 value), runs never merged or split)
  serialize                 <- DataChunk.Serialize, DataChunk.cs:173-206: tries the serializers in the
                               order {Raw, RLE} (:165-168) and keeps the first STRICTLY smaller one
  deserialize               <- DataChunk.DeserializeData (DataChunk.cs:208-235) +
                               DataChunkSerializer.cs:36-46,102-142; the Viewer's decoder
                               (DistributedMandelbrotViewer.py:35-60) reads the same format
"""
from __future__ import annotations

import numpy as np

RAW_CODE, RLE_CODE = 0x00, 0x01


def rle_runs(data: np.ndarray):
    d = np.ascontiguousarray(data, dtype=np.uint8).ravel()
    if d.size == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.uint8)
    starts = np.flatnonzero(np.concatenate(([True], d[1:] != d[:-1])))
    lengths = np.diff(np.concatenate((starts, [d.size])))
    return lengths, d[starts]


def rle_encode(data: np.ndarray) -> bytes:
    lengths, values = rle_runs(data)
    rec = np.zeros(len(lengths), dtype=np.dtype([("len", "<u4"), ("val", "u1")]))
    rec["len"] = lengths
    rec["val"] = values
    return bytes([RLE_CODE]) + rec.tobytes()


def raw_encode(data: np.ndarray) -> bytes:
    return bytes([RAW_CODE]) + np.ascontiguousarray(data, dtype=np.uint8).ravel().tobytes()


def serialize(data: np.ndarray) -> bytes:
    best = raw_encode(data)
    rle = rle_encode(data)
    if len(rle) < len(best):
        best = rle
    return best


def deserialize(stream: bytes, size: int) -> np.ndarray:
    code = stream[0]
    if code == RAW_CODE:
        return np.frombuffer(stream, np.uint8, count=size, offset=1).copy()
    if code != RLE_CODE:
        raise ValueError("No serializer found for chunk file")
    out = np.empty(size, np.uint8)
    pos, off = 0, 1
    while pos < size:
        run = int.from_bytes(stream[off:off + 4], "little")
        val = stream[off + 4]
        off += 5
        if run == 0:
            raise ValueError("Encountered run of length 0")
        if pos + run > size:
            raise ValueError("Data exceeds chunk expected length")
        out[pos:pos + run] = val
        pos += run
    return out
