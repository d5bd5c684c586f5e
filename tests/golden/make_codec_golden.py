#!/usr/bin/env python3
"""Generate tests/golden/codec_vectors.npz by EXECUTING THE REFERENCE'S OWN READ PATH.

Run once, in the build container (needs /root/reference; the GPU box does not have it):

    python tests/golden/make_codec_golden.py

The reference's Viewer (/root/reference/DistributedMandelbrotViewer/DistributedMandelbrotViewer.py, "Viewer.py") is plain
Python: `deserialize_rle` / `chunk_data_to_value_array` (:35-60) decode the stream DataChunk.Serialize writes
(DataChunk.cs:173-206, DataChunkSerializer.cs:56-100) and `get_chunk` (:62-108) speaks the DataServer protocol
(DataServer.cs:156-224).  This script imports that file unmodified (matplotlib is only imported, never shown) and

  (a) feeds `chunk_data_to_value_array` the serialised streams of the six golden 4096^2 tiles (the tiles are the
      reference's own `process_workload` output -- reference_vectors.npz holds their hashes; the C oracle reproduces
      them bit for bit and is checked against those hashes here before its bytes are used) and of adversarial run
      patterns, and stores hash(stream), its length and codec, and hash(what the reference decoded);
  (b) runs the reference's `get_chunk` against this repository's DataServer stand-in over a ChunkStore holding a Regular,
      a Never and an Immediate chunk, and stores what it returned.

So the codec vectors pin "the stream that the reference's own decoder turns back into the reference's own tile".  What
they cannot pin is the C# *encoder* byte for byte (no dotnet here) -- but for a given tile the stream is unique: Raw is
the bytes themselves, and RLE runs are maximal by construction (DataChunkSerializer.cs:66-90 never splits or merges a
run), so a stream that decodes to the tile, has maximal runs and follows the Raw-unless-RLE-is-strictly-smaller rule
(DataChunk.cs:186-196) IS the C# stream.  The tests check all three properties.

Nothing from the reference is copied into this repository: the module is loaded from where it lies.
"""
from __future__ import annotations

import hashlib
import importlib.util
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_VIEWER = "/root/reference/DistributedMandelbrotViewer/DistributedMandelbrotViewer.py"
OUT = os.path.join(HERE, "codec_vectors.npz")
CHUNK = 4096 * 4096


def load_reference_viewer():
    os.environ.setdefault("MPLBACKEND", "Agg")     # the module imports pyplot at the top; nothing is ever shown
    spec = importlib.util.spec_from_file_location("reference_viewer", REF_VIEWER)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def pattern(name: str) -> np.ndarray:
    """Adversarial byte patterns, regenerated from their name alone (the tests call this too)."""
    rs = np.random.RandomState(abs(hash_name(name)) % (2 ** 31))
    if name == "all_zero_chunk":            # one run of 2^24: the "Never" chunk as the DataServer sends it
        return np.zeros(CHUNK, np.uint8)
    if name == "all_one_chunk":
        return np.ones(CHUNK, np.uint8)
    if name == "alternating_chunk":         # runs of length 1: RLE would be 5x the size -> Raw
        return (np.arange(CHUNK) & 1).astype(np.uint8)
    if name == "long_runs_chunk":           # run lengths around the 2^8 / 2^16 boundaries of the u32 length field
        lens = [1, 255, 256, 257, 65535, 65536, 65537, 1, 1, 2, 1 << 20, 3]
        vals = [255, 0, 1, 254, 7, 0, 9, 9 ^ 1, 0, 1, 128, 2]
        rest = CHUNK - sum(lens)
        more = []
        while rest > 0:
            k = int(min(rest, rs.randint(1, 1 << 18)))
            more.append(k)
            rest -= k
        lens += more
        vals += [int(v) for v in rs.randint(0, 256, len(more))]
        for i in range(1, len(vals)):       # neighbouring runs must differ, or they would not be runs
            if vals[i] == vals[i - 1]:
                vals[i] = (vals[i] + 1) % 256
        return np.repeat(np.array(vals, np.uint8), np.array(lens, np.int64))
    if name == "noisy_chunk":               # every byte random: Raw
        return rs.randint(0, 256, CHUNK, dtype=np.uint8)
    if name == "break_even_chunk":          # runs of 5 with one odd run: RLE one byte LARGER than Raw -> Raw
        d = np.repeat((np.arange(CHUNK // 5 + 1) % 251).astype(np.uint8), 5)[:CHUNK]
        return d
    if name == "rle_wins_by_a_hair_chunk":  # runs of 5, two of them merged by equal values: RLE strictly smaller
        v = (np.arange(CHUNK // 5 + 1) % 251).astype(np.uint8)
        v[1] = v[0]
        v[3] = v[2]
        d = np.repeat(v, 5)[:CHUNK]
        return d
    if name == "tie_small":                 # raw 11 bytes, rle 11 bytes: the first serializer (Raw) stays
        return np.array([1] * 5 + [2] * 5, np.uint8)
    if name == "single_byte":
        return np.array([200], np.uint8)
    raise KeyError(name)


def hash_name(name: str) -> int:
    return int.from_bytes(hashlib.sha256(name.encode()).digest()[:4], "little")


PATTERNS = ["all_zero_chunk", "all_one_chunk", "alternating_chunk", "long_runs_chunk", "noisy_chunk", "break_even_chunk",
            "rle_wins_by_a_hair_chunk", "tie_small", "single_byte"]


def sha(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()


def main():
    from oracle.oracle import COracle
    from oracle.serializer import serialize
    viewer = load_reference_viewer()
    golden = np.load(os.path.join(HERE, "reference_vectors.npz"))
    oracle = COracle()
    data = {}

    def record(prefix, tile_bytes):
        stream = serialize(tile_bytes)
        decoded = viewer.chunk_data_to_value_array(bytearray(stream))      # the reference's decoder, Viewer.py:48-60
        assert bytes(decoded) == tile_bytes.tobytes(), prefix
        data[f"{prefix}/stream_sha256"] = np.array(sha(stream))
        data[f"{prefix}/stream_len"] = np.array(len(stream))
        data[f"{prefix}/codec"] = np.array(stream[0])
        data[f"{prefix}/decoded_sha256"] = np.array(sha(decoded))
        print(prefix, "codec", stream[0], "len", len(stream), sha(stream)[:16])

    for key in golden["full/names"]:
        level, mrd, ir, ii = (int(x) for x in golden[f"full/{key}/params"])
        byts = oracle.datachunk(level, mrd, ir, ii, want_counts=False)[1].ravel()
        # the tile must BE the reference's process_workload output before its stream may be pinned
        assert sha(byts.tobytes()) == str(golden[f"full/{key}/bytes_sha256"]), key
        record(f"tile/{key}", byts)
    data["tile/names"] = golden["full/names"]
    for name in PATTERNS:
        record(f"pattern/{name}", pattern(name))
    data["pattern/names"] = np.array(PATTERNS)

    # (b) the reference's get_chunk against the DataServer stand-in
    from distributedmandelbrot_amd.chunkstore import ChunkStore
    from distributedmandelbrot_amd.server import DataServer
    with tempfile.TemporaryDirectory() as tmp:
        store = ChunkStore(tmp)
        store.save_chunk(7, 1, 2, pattern("long_runs_chunk"))      # Regular (RLE file)
        store.save_chunk(7, 0, 0, pattern("all_zero_chunk"))       # Never: index entry only
        store.save_chunk(7, 6, 6, pattern("all_one_chunk"))        # Immediate
        store.save_chunk(7, 3, 3, pattern("noisy_chunk"))          # Regular (Raw file)
        with DataServer(store) as ds:
            for (ir, ii), name in {(1, 2): "long_runs_chunk", (0, 0): "all_zero_chunk", (6, 6): "all_one_chunk",
                                   (3, 3): "noisy_chunk"}.items():
                vs, ok = viewer.get_chunk("127.0.0.1", ds.port, 7, ir, ii)          # Viewer.py:62-108
                assert ok and vs.dtype == np.uint8 and np.array_equal(vs, pattern(name)), name
                data[f"get_chunk/{name}/sha256"] = np.array(sha(vs.tobytes()))
            assert viewer.get_chunk("127.0.0.1", ds.port, 7, 5, 5) == (None, False)  # not available (0x02)
            try:
                viewer.get_chunk("127.0.0.1", ds.port, 7, 7, 0)                      # index >= level -> 0x01
                raise AssertionError("the reference's get_chunk accepted a rejected request")
            except Exception as e:   # noqa: BLE001 -- the reference raises a bare Exception
                assert "rejected" in str(e)
    data["meta/reference_viewer_sha256"] = np.array(sha(open(REF_VIEWER, "rb").read()))
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
