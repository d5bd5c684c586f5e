"""Chunk codec tests: the CPU restatement of DataChunkSerializer/DataChunk.Serialize,
and (on the GPU box) the on-device serialiser against it."""
import numpy as np
import pytest

from oracle.serializer import RAW_CODE, RLE_CODE, deserialize, raw_encode, rle_encode, rle_runs, serialize


def test_rle_hand_example():
    data = np.array([7, 7, 7, 0, 0, 9], np.uint8)
    # (u32 3, 7)(u32 2, 0)(u32 1, 9) after the 0x01 code byte -- DataChunkSerializer.cs:56-100
    assert rle_encode(data) == bytes([1, 3, 0, 0, 0, 7, 2, 0, 0, 0, 0, 1, 0, 0, 0, 9])
    assert raw_encode(data) == bytes([0, 7, 7, 7, 0, 0, 9])
    assert serialize(data) == raw_encode(data)                      # 16 bytes RLE vs 7 raw
    assert serialize(np.zeros(100, np.uint8)) == bytes([1, 100, 0, 0, 0, 0])
    # tie -> Raw (first serializer wins unless a later one is strictly smaller, DataChunk.cs:190)
    tie = np.array([1, 1, 1, 1, 1, 2, 2, 2, 2, 2], np.uint8)       # raw 11, rle 11
    assert serialize(tie)[0] == RAW_CODE


def test_rle_roundtrip_random():
    rs = np.random.RandomState(3)
    for n, p in [(1, 0.5), (1000, 0.01), (5000, 0.5), (65536, 0.001)]:
        data = np.cumsum(rs.rand(n) < p).astype(np.uint8)
        for enc in (rle_encode(data), raw_encode(data), serialize(data)):
            assert np.array_equal(deserialize(enc, n), data)
        lengths, values = rle_runs(data)
        assert lengths.sum() == n and (np.diff(values.astype(int)) != 0).all()
    with pytest.raises(ValueError):
        deserialize(bytes([1, 0, 0, 0, 0, 5]), 4)                     # run of length 0
    with pytest.raises(ValueError):
        deserialize(bytes([1, 9, 0, 0, 0, 5]), 4)                     # run exceeds the chunk
    with pytest.raises(ValueError):
        deserialize(bytes([7, 1, 2, 3]), 3)                           # unknown code


@pytest.mark.gpu
def test_gpu_serialiser_matches_reference_codec(gpu, oracle):
    from distributedmandelbrot_amd import View
    cases = [
        ("datachunk", (4, 256, 0, 0)),       # exterior: long runs -> RLE
        ("datachunk", (20, 1024, 9, 10)),    # inside the cardioid: all bytes 0 -> one run
        ("datachunk", (4, 256, 1, 2)),       # boundary tile
        ("view", (View(-2.0, -1.5, 3.0, 3.0, 333, 217), 100)),
        ("view", (View(-0.755, 0.10, 0.02, 0.02, 512, 512), 1024)),   # noisy: Raw wins
        ("view", (View(1.0, 1.0, 1.0, 1.0, 1, 1), 10)),
    ]
    for kind, arg in cases:
        if kind == "datachunk":
            byts, _, st = gpu.datachunk(*arg)
        else:
            _, byts, st = gpu.compute_view(arg[0], arg[1], want_counts=False)
        flat = byts.ravel()
        want = serialize(flat)
        got, codec = gpu.serialize_last()
        assert st.rle_runs == len(rle_runs(flat)[0]), (kind, arg)
        assert codec == want[0] and got == want, (kind, arg, codec, len(got), len(want))
        assert np.array_equal(deserialize(got, flat.size), flat)
    # the all-in-set tile is what the server stores as a "Never" index entry (DataStorage.cs:74-75)
    byts, _, st = gpu.datachunk(20, 1024, 9, 10)
    assert st.all_bytes_zero and st.rle_runs == 1 and gpu.serialize_last() == (bytes([RLE_CODE, 0, 0, 0, 1, 0]), RLE_CODE)
