"""The chunk codec and the DataServer wire, pinned by the reference's own read path.

tests/golden/codec_vectors.npz was produced by tests/golden/make_codec_golden.py, which EXECUTES the reference's
Viewer (DistributedMandelbrotViewer.py:35-108: deserialize_rle, chunk_data_to_value_array, get_chunk) on the serialised
streams of the six golden tiles and of adversarial run patterns: a pinned stream hash therefore means "the stream the
reference's decoder turns back into the reference's own tile".  Here
  * the CPU suite checks that the codec restatement (oracle/serializer.py) AND the product's host codec
    (chunkstore.serialize_chunk) reproduce every pinned stream, that the runs are maximal and the Raw/RLE choice
    follows DataChunk.cs:186-196 -- which together make the stream the unique one the C# encoder writes;
  * where /root/reference exists (the build container) the reference's get_chunk is run live against the DataServer
    stand-in over a ChunkStore;
  * the GPU suite asserts that the on-device serialiser (mbk_serialize_last) emits exactly the pinned streams for the
    six golden tiles."""
import hashlib
import importlib.util
import os
import struct

import numpy as np
import pytest

from conftest import ROOT
from distributedmandelbrot_amd.chunkstore import ChunkStore, deserialize_chunk, serialize_chunk
from distributedmandelbrot_amd.server import DataServer
from oracle.serializer import deserialize, rle_runs, serialize

REF_VIEWER = "/root/reference/DistributedMandelbrotViewer/DistributedMandelbrotViewer.py"


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def codec_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "codec_vectors.npz"))


@pytest.fixture(scope="module")
def gen():
    """The pattern generator of the golden script (patterns are regenerated from their names, not stored)."""
    return _load(os.path.join(ROOT, "tests", "golden", "make_codec_golden.py"), "make_codec_golden")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def check_stream_is_the_csharp_stream(stream: bytes, data: np.ndarray):
    """Unique-stream argument: decodes to `data`; RLE records are maximal runs (DataChunkSerializer.cs:66-90 starts a new
    record only when the value changes); Raw unless RLE is STRICTLY smaller (DataChunk.cs:186-196, serializers in the
    order Raw, RLE :165-168)."""
    n = data.size
    lengths, values = rle_runs(data)
    rle_size, raw_size = 1 + 5 * len(lengths), 1 + n
    if stream[0] == 0x01:
        assert rle_size < raw_size and len(stream) == rle_size
        rec = np.frombuffer(stream, dtype=np.dtype([("len", "<u4"), ("val", "u1")]), offset=1)
        assert np.array_equal(rec["len"], lengths) and np.array_equal(rec["val"], values)
        assert (np.diff(rec["val"].astype(int)) != 0).all() and (rec["len"] > 0).all()
    else:
        assert stream[0] == 0x00 and rle_size >= raw_size and stream[1:] == data.tobytes()


def test_patterns_reproduce_the_streams_the_reference_decoder_accepted(codec_golden, gen):
    for name in codec_golden["pattern/names"]:
        data = gen.pattern(str(name))
        for enc in (serialize, serialize_chunk):        # the checker's codec and the product's host codec
            stream = enc(data)
            assert sha(stream) == str(codec_golden[f"pattern/{name}/stream_sha256"]), (name, enc.__name__)
            assert len(stream) == int(codec_golden[f"pattern/{name}/stream_len"]) and stream[0] == int(codec_golden[f"pattern/{name}/codec"])
        check_stream_is_the_csharp_stream(stream, data)
        assert sha(deserialize(stream, data.size)) == str(codec_golden[f"pattern/{name}/decoded_sha256"]) == sha(data)
        if data.size == 4096 * 4096:
            assert np.array_equal(deserialize_chunk(stream), data)
    # the Never / Immediate chunks as the DataServer sends them (CreateNeverChunk / CreateImmediateChunk -> one run)
    assert serialize(gen.pattern("all_zero_chunk")) == bytes([1]) + struct.pack("<IB", 1 << 24, 0)
    assert serialize(gen.pattern("all_one_chunk")) == bytes([1]) + struct.pack("<IB", 1 << 24, 1)


def test_golden_tiles_reproduce_the_streams_the_reference_decoder_accepted(codec_golden, golden, oracle):
    for key in codec_golden["tile/names"]:
        level, mrd, ir, ii = (int(x) for x in golden[f"full/{key}/params"])
        byts = oracle.datachunk(level, mrd, ir, ii, want_counts=False)[1].ravel()
        assert sha(byts.tobytes()) == str(golden[f"full/{key}/bytes_sha256"])        # the reference's own tile
        stream = serialize_chunk(byts)
        assert sha(stream) == str(codec_golden[f"tile/{key}/stream_sha256"]), key
        assert len(stream) == int(codec_golden[f"tile/{key}/stream_len"])
        check_stream_is_the_csharp_stream(stream, byts)
        assert str(codec_golden[f"tile/{key}/decoded_sha256"]) == str(golden[f"full/{key}/bytes_sha256"])


@pytest.mark.skipif(not os.path.exists(REF_VIEWER), reason="the reference tree is not on this machine (GPU box)")
def test_reference_viewer_reads_the_dataserver_stand_in_live(tmp_path, codec_golden, gen):
    """The reference's unmodified get_chunk / chunk_data_to_value_array (Viewer.py:35-108) against server.DataServer over
    a ChunkStore: Regular (RLE and Raw files), Never, Immediate, not-available and rejected requests."""
    os.environ.setdefault("MPLBACKEND", "Agg")
    viewer = _load(REF_VIEWER, "reference_viewer_live")
    assert sha(open(REF_VIEWER, "rb").read()) == str(codec_golden["meta/reference_viewer_sha256"])
    store = ChunkStore(str(tmp_path))
    placed = {(1, 2): "long_runs_chunk", (0, 0): "all_zero_chunk", (4, 4): "all_one_chunk", (3, 3): "noisy_chunk"}
    for (ir, ii), name in placed.items():
        store.save_chunk(5, ir, ii, gen.pattern(name))
    with DataServer(store) as ds:
        for (ir, ii), name in placed.items():
            vs, ok = viewer.get_chunk("127.0.0.1", ds.port, 5, ir, ii)
            assert ok and vs.dtype == np.uint8 and sha(vs.tobytes()) == str(codec_golden[f"get_chunk/{name}/sha256"])
        assert viewer.get_chunk("127.0.0.1", ds.port, 5, 2, 2) == (None, False)
        with pytest.raises(Exception, match="rejected"):
            viewer.get_chunk("127.0.0.1", ds.port, 5, 5, 0)
    # and the decoder on a stream the product's host codec makes right now
    stream = serialize_chunk(gen.pattern("rle_wins_by_a_hair_chunk")[: 5 * 4000])
    assert bytes(viewer.chunk_data_to_value_array(bytearray(stream))) == gen.pattern("rle_wins_by_a_hair_chunk")[: 5 * 4000].tobytes()


@pytest.mark.gpu
def test_gpu_serialiser_emits_the_pinned_streams(gpu, codec_golden, golden):
    """mbk_serialize_last for the six golden tiles == the streams the reference's decoder turned back into the
    reference's own tiles (hash, length, codec)."""
    for key in codec_golden["tile/names"]:
        level, mrd, ir, ii = (int(x) for x in golden[f"full/{key}/params"])
        byts, _, st = gpu.datachunk(level, mrd, ir, ii)
        assert sha(byts.tobytes()) == str(golden[f"full/{key}/bytes_sha256"]), key
        stream, codec = gpu.serialize_last()
        assert sha(stream) == str(codec_golden[f"tile/{key}/stream_sha256"]), key
        assert len(stream) == int(codec_golden[f"tile/{key}/stream_len"]) and codec == int(codec_golden[f"tile/{key}/codec"])
        assert st.rle_runs * 5 + 1 == len(stream) or codec == 0
