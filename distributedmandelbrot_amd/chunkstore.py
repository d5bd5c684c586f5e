"""On-disk chunk store in the reference's format (DataStorage.cs), so that tiles
computed here can be served by the untouched C# DataServer / read by the untouched Viewer.

    <parent>/Data/_index.dat    concatenated entries: u32 level, u32 indexReal, u32 indexImag, i32 type
                                (0 Regular, 1 Never, 2 Immediate -- enum order DataStorage.cs:41-49; the
                                code writes 4 bytes, :373-374, although the header comment says uint8),
                                and for Regular only: i32 filenameLength, ASCII filename (:376-385)
    <parent>/Data/<filename>    Regular chunks: u8 codec + payload, exactly DataChunk.Serialize
                                (DataChunk.cs:173-206); name "{level};{indexReal};{indexImag}" plus a
                                numeric suffix if it exists already (DataStorage.cs:392-405)

Never (all bytes 0) and Immediate (all bytes 1) chunks are index-only (DataStorage.cs:71-84,424-425).
Host-side code; the serialised stream can come from the GPU (`MandelbrotDevice.serialize_last`).
"""
from __future__ import annotations

import os
import struct
import threading
from dataclasses import dataclass
from typing import Iterator, Optional

import numpy as np

CHUNK_BYTES = 4096 * 4096
TYPE_REGULAR, TYPE_NEVER, TYPE_IMMEDIATE = 0, 1, 2
RAW_CODE, RLE_CODE = 0x00, 0x01


@dataclass(frozen=True)
class IndexEntry:
    level: int
    index_real: int
    index_imag: int
    type: int
    filename: str = ""


def rle_payload(data: np.ndarray) -> bytes:
    """DataChunkSerializer.cs:56-100: repeated (u32 runLength little-endian, u8 value)."""
    d = np.ascontiguousarray(data, dtype=np.uint8).ravel()
    starts = np.flatnonzero(np.concatenate(([True], d[1:] != d[:-1])))
    rec = np.zeros(len(starts), dtype=np.dtype([("len", "<u4"), ("val", "u1")]))
    rec["len"] = np.diff(np.concatenate((starts, [d.size])))
    rec["val"] = d[starts]
    return rec.tobytes()


def serialize_chunk(data: np.ndarray) -> bytes:
    """DataChunk.Serialize (DataChunk.cs:173-206): Raw unless RLE is strictly shorter."""
    d = np.ascontiguousarray(data, dtype=np.uint8).ravel()
    rle = rle_payload(d)
    if 1 + len(rle) < 1 + d.size:
        return bytes([RLE_CODE]) + rle
    return bytes([RAW_CODE]) + d.tobytes()


def deserialize_chunk(stream: bytes, size: int = CHUNK_BYTES) -> np.ndarray:
    """DataChunk.DeserializeData (DataChunk.cs:208-235) / DataChunkSerializer.cs:36-46,102-142."""
    code = stream[0]
    if code == RAW_CODE:
        return np.frombuffer(stream, np.uint8, count=size, offset=1).copy()
    if code != RLE_CODE:
        raise ValueError("No serializer found for chunk file")
    rec = np.frombuffer(stream, dtype=np.dtype([("len", "<u4"), ("val", "u1")]), offset=1)
    if (rec["len"] == 0).any():
        raise ValueError("Encountered run of length 0")
    total = int(rec["len"].astype(np.int64).sum())
    if total != size:
        raise ValueError("Data exceeds chunk expected length" if total > size else "Chunk data too short")
    return np.repeat(rec["val"], rec["len"].astype(np.int64))


class ChunkStore:
    def __init__(self, parent_dir: str):
        self.data_dir = os.path.join(parent_dir, "Data")          # DataStorage.cs:15-16
        self.index_path = os.path.join(self.data_dir, "_index.dat")  # :18-20
        self._lock = threading.Lock()
        os.makedirs(self.data_dir, exist_ok=True)                  # SetUpDataDirectoryIfNeeded, :129-146
        if not os.path.exists(self.index_path):
            open(self.index_path, "wb").close()

    # -- index ----------------------------------------------------------------------------
    @staticmethod
    def _pack_entry(e: IndexEntry) -> bytes:
        out = struct.pack("<IIIi", e.level, e.index_real, e.index_imag, e.type)
        if e.type == TYPE_REGULAR:
            name = e.filename.encode("ascii")
            out += struct.pack("<i", len(name)) + name
        return out

    def entries(self) -> Iterator[IndexEntry]:
        """GetIndexEntriesEnumerator (DataStorage.cs:294-322)."""
        with self._lock:
            raw = open(self.index_path, "rb").read()
        off = 0
        while off < len(raw):
            level, ir, ii, typ = struct.unpack_from("<IIIi", raw, off)
            off += 16
            name = ""
            if typ == TYPE_REGULAR:
                (n,) = struct.unpack_from("<i", raw, off)
                off += 4
                name = raw[off:off + n].decode("ascii")
                off += n
            yield IndexEntry(level, ir, ii, typ, name)

    def _generate_filename(self, level: int, ir: int, ii: int) -> str:
        base = f"{level};{ir};{ii}"                                # DataStorage.cs:392-405
        if not os.path.exists(os.path.join(self.data_dir, base)):
            return base
        k = 0
        while os.path.exists(os.path.join(self.data_dir, base + str(k))):
            k += 1
        return base + str(k)

    # -- save -----------------------------------------------------------------------------
    def _append(self, entry: IndexEntry, file_bytes: Optional[bytes]) -> IndexEntry:
        with self._lock:
            if entry.type == TYPE_REGULAR:
                entry = IndexEntry(entry.level, entry.index_real, entry.index_imag, TYPE_REGULAR,
                                   self._generate_filename(entry.level, entry.index_real, entry.index_imag))
                with open(os.path.join(self.data_dir, entry.filename), "wb") as f:
                    f.write(file_bytes)
            with open(self.index_path, "ab") as f:
                f.write(self._pack_entry(entry))
        return entry

    def save_chunk(self, level: int, index_real: int, index_imag: int, data: np.ndarray) -> IndexEntry:
        """SaveDataChunk (DataStorage.cs:410-427) for raw tile bytes (what a worker sent)."""
        d = np.ascontiguousarray(data, dtype=np.uint8).ravel()
        if d.size != CHUNK_BYTES:
            raise ValueError("a DataChunk has 16 777 216 bytes")
        if not d.any():                                            # IsNeverChunk, DataChunk.cs:82
            return self._append(IndexEntry(level, index_real, index_imag, TYPE_NEVER), None)
        if (d == 1).all():                                         # IsImmediateChunk, DataChunk.cs:87
            return self._append(IndexEntry(level, index_real, index_imag, TYPE_IMMEDIATE), None)
        return self._append(IndexEntry(level, index_real, index_imag, TYPE_REGULAR), serialize_chunk(d))

    def save_from_device(self, dev, level: int, mrd: int, index_real: int, index_imag: int) -> IndexEntry:
        """Compute a tile on the GPU and store it WITHOUT a host pass over the pixels: the all-0 / all-1
        flags and the Raw/RLE stream come from the device (mbk_stats, mbk_serialize_last)."""
        _, _, st = dev.datachunk(level, mrd, index_real, index_imag)
        if st.all_bytes_zero:
            return self._append(IndexEntry(level, index_real, index_imag, TYPE_NEVER), None)
        if st.all_bytes_one:
            return self._append(IndexEntry(level, index_real, index_imag, TYPE_IMMEDIATE), None)
        stream, _ = dev.serialize_last()
        return self._append(IndexEntry(level, index_real, index_imag, TYPE_REGULAR), stream)

    # -- load -----------------------------------------------------------------------------
    def find(self, level: int, index_real: int, index_imag: int) -> Optional[IndexEntry]:
        """First matching index entry (TryLoadChunks scans the index linearly, DataStorage.cs:256-292)."""
        for e in self.entries():
            if (e.level, e.index_real, e.index_imag) == (level, index_real, index_imag):
                return e
        return None

    def load_serialized(self, entry: IndexEntry) -> bytes:
        """The stream DataChunk.Serialize would produce for this entry (what the DataServer sends)."""
        if entry.type == TYPE_REGULAR:
            with open(os.path.join(self.data_dir, entry.filename), "rb") as f:
                return f.read()
        value = 0 if entry.type == TYPE_NEVER else 1              # CreateNeverChunk / CreateImmediateChunk
        return bytes([RLE_CODE]) + struct.pack("<IB", CHUNK_BYTES, value)

    def load_chunk(self, level: int, index_real: int, index_imag: int) -> Optional[np.ndarray]:
        e = self.find(level, index_real, index_imag)
        if e is None:
            return None
        return deserialize_chunk(self.load_serialized(e))

    def completed(self):
        """(level, indexReal, indexImag) of every stored chunk: what the Distributer reloads at start
        (Distributer.cs:124,165-175)."""
        return {(e.level, e.index_real, e.index_imag) for e in self.entries()}
