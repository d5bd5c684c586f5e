"""Tests of the server-side stand-ins: the on-disk format of DataStorage.cs, the
Distributer lease semantics and the DataServer->Viewer protocol.  The Viewer client below restates
DistributedMandelbrotViewer.py:35-108 (test infrastructure)."""
import os
import socket
import struct
import threading
import time

import numpy as np
import pytest

from distributedmandelbrot_amd import worker
from distributedmandelbrot_amd.chunkstore import (CHUNK_BYTES, TYPE_IMMEDIATE, TYPE_NEVER, TYPE_REGULAR,
                                                  ChunkStore, deserialize_chunk, serialize_chunk)
from distributedmandelbrot_amd.server import DataServer, Distributer
from oracle.serializer import serialize as oracle_serialize

QUIET = lambda *a: None  # noqa: E731


def viewer_get_chunk(addr, port, level, index_real, index_imag):
    """Viewer.py:62-108 get_chunk + :35-60 decoders."""
    with socket.create_connection((addr, port)) as s:
        s.sendall(struct.pack("III", level, index_real, index_imag))
        status = s.recv(1)[0]
        if status == 0x02:
            return None, False
        if status == 0x01:
            raise Exception("Request was rejected")
        if status != 0x00:
            raise Exception("Unknown request status code: " + str(status))
        n = struct.unpack("I", s.recv(4))[0]
        raw = bytearray()
        while len(raw) < n:
            raw += s.recv(n - len(raw))
    code, payload = raw[0], bytes(raw[1:])
    if code == 0x00:
        data = np.frombuffer(payload, np.uint8)
    elif code == 0x01:
        out = bytearray()
        for i in range(0, len(payload), 5):   # Viewer.py:43-46: unpack("IB")
            run, val = struct.unpack("<IB", payload[i:i + 5])
            out += bytes([val]) * run
        data = np.frombuffer(bytes(out), np.uint8)
    else:
        raise Exception("Unknown serialization type code")
    assert len(data) == CHUNK_BYTES
    return data, True


def tile(seed):
    rs = np.random.RandomState(seed)
    t = np.repeat(rs.randint(0, 256, CHUNK_BYTES // 4096, dtype=np.uint8), 4096)  # long runs -> RLE
    return t


def test_index_and_file_format_bytes(tmp_path):
    store = ChunkStore(str(tmp_path))
    never = np.zeros(CHUNK_BYTES, np.uint8)
    imm = np.ones(CHUNK_BYTES, np.uint8)
    reg = tile(1)
    assert store.save_chunk(4, 0, 1, never).type == TYPE_NEVER
    assert store.save_chunk(4, 2, 3, imm).type == TYPE_IMMEDIATE
    e = store.save_chunk(10, 3, 5, reg)
    assert e.type == TYPE_REGULAR and e.filename == "10;3;5"
    e2 = store.save_chunk(10, 3, 5, reg)            # same tile again -> numeric suffix (DataStorage.cs:392-405)
    assert e2.filename == "10;3;50"
    raw = open(os.path.join(tmp_path, "Data", "_index.dat"), "rb").read()
    want = (struct.pack("<IIIi", 4, 0, 1, 1) + struct.pack("<IIIi", 4, 2, 3, 2)
            + struct.pack("<IIIii", 10, 3, 5, 0, 6) + b"10;3;5" + struct.pack("<IIIii", 10, 3, 5, 0, 7) + b"10;3;50")
    assert raw == want
    on_disk = open(os.path.join(tmp_path, "Data", "10;3;5"), "rb").read()
    assert on_disk == oracle_serialize(reg) == serialize_chunk(reg) and on_disk[0] == 0x01
    assert sorted(os.listdir(os.path.join(tmp_path, "Data"))) == ["10;3;5", "10;3;50", "_index.dat"]
    # read back
    assert np.array_equal(store.load_chunk(10, 3, 5), reg)
    assert not store.load_chunk(4, 0, 1).any() and (store.load_chunk(4, 2, 3) == 1).all()
    assert store.load_chunk(4, 1, 1) is None
    assert store.completed() == {(4, 0, 1), (4, 2, 3), (10, 3, 5)}
    noisy = np.random.RandomState(2).randint(0, 256, CHUNK_BYTES, dtype=np.uint8)
    assert serialize_chunk(noisy)[0] == 0x00 and np.array_equal(deserialize_chunk(serialize_chunk(noisy)), noisy)


def test_distributer_to_store_to_dataserver_to_viewer(tmp_path):
    store = ChunkStore(str(tmp_path))
    tiles = {}

    def compute(level, mrd, ir, ii):
        t = tile(level * 100 + ir * 10 + ii) if (ir, ii) != (0, 0) else np.zeros(CHUNK_BYTES, np.uint8)
        tiles[(level, ir, ii)] = t
        return t

    with Distributer([(2, 64)], store=store) as dist, DataServer(store) as ds:
        done = worker.run_farm("127.0.0.1", dist.port, devices=[0, 1, 2], make_compute=lambda d: compute, log=QUIET)
        assert sum(done) == 4
        for _ in range(200):
            if dist.received == 4 and dist.all_done():
                break
            threading.Event().wait(0.02)
        assert dist.all_done() and not dist.rejected
        for (level, ir, ii), t in tiles.items():
            data, ok = viewer_get_chunk("127.0.0.1", ds.port, level, ir, ii)
            assert ok and np.array_equal(data, t)
        assert viewer_get_chunk("127.0.0.1", ds.port, 3, 0, 0) == (None, False)      # not computed
        with pytest.raises(Exception, match="rejected"):
            viewer_get_chunk("127.0.0.1", ds.port, 2, 2, 0)                           # index >= level
    # the all-zero tile became a Never entry without a file (DataStorage.cs:74-75,424-425)
    assert store.find(2, 0, 0).type == TYPE_NEVER and "2;0;0" not in os.listdir(os.path.join(tmp_path, "Data"))
    # a restarted Distributer reloads the completed set from the index and has nothing left to hand out
    with Distributer([(2, 64)], store=store) as dist2:
        assert worker.request_workload("127.0.0.1", dist2.port) is None


def test_distributer_lease_expiry_rejects_late_result():
    with Distributer([(1, 8)], lease_seconds=0.05) as dist:
        w = worker.request_workload("127.0.0.1", dist.port)
        assert w == (1, 8, 0, 0)
        threading.Event().wait(0.1)                                   # lease expires (Distributer.cs:22)
        assert worker.submit_workload("127.0.0.1", dist.port, w, np.zeros(CHUNK_BYTES, np.uint8)) is False
        assert dist.rejected == [w]
        assert worker.request_workload("127.0.0.1", dist.port) == w   # re-offered


@pytest.mark.gpu
def test_gpu_tiles_through_the_whole_pipeline(tmp_path, gpu, oracle):
    """HIP worker -> Distributer stand-in -> reference disk format -> DataServer stand-in -> Viewer decoder,
    and the direct device->store path (stats + on-device serialiser), against the CPU oracle."""
    store = ChunkStore(str(tmp_path))
    with Distributer([(3, 128)], store=store) as dist, DataServer(store) as ds:
        done = worker.run_farm("127.0.0.1", dist.port, devices=[0, 0], log=QUIET)
        assert sum(done) == 9
        for _ in range(500):
            if dist.received == 9:
                break
            threading.Event().wait(0.02)
        for ir, ii in [(0, 0), (1, 1), (2, 1)]:
            data, ok = viewer_get_chunk("127.0.0.1", ds.port, 3, ir, ii)
            want = oracle.datachunk(3, 128, ir, ii, want_counts=False)[1].ravel()
            assert ok and np.array_equal(data, want)
    direct = ChunkStore(str(tmp_path / "direct"))
    for level, mrd, ir, ii in [(20, 64, 9, 10), (4, 256, 0, 0), (3, 128, 1, 1)]:
        e = direct.save_from_device(gpu, level, mrd, ir, ii)
        want = oracle.datachunk(level, mrd, ir, ii, want_counts=False)[1].ravel()
        assert np.array_equal(direct.load_chunk(level, ir, ii), want)
        if not want.any():
            assert e.type == TYPE_NEVER


def test_duplicate_response_for_one_lease_is_rejected_while_first_is_arriving():
    """Two responses for the same leased tile (an expired worker and its successor, or a retry): the
    reference handles connections one at a time and accepts only the first (Distributer.cs:404-423); the
    threaded stand-in must claim the lease atomically -- the second gets 0x21 even while the first
    payload is still on the wire, and a failed transfer makes the lease live again."""
    import socket
    import struct
    with Distributer([(1, 8)]) as dist:
        w = worker.request_workload("127.0.0.1", dist.port)
        a = socket.create_connection(("127.0.0.1", dist.port))
        a.sendall(struct.pack("<BIIII", 0x01, *w))
        assert a.recv(1) == bytes([0x20])
        a.sendall(bytes(1000))                                   # payload in progress ...
        assert worker.submit_workload("127.0.0.1", dist.port, w, np.zeros(CHUNK_BYTES, np.uint8)) is False
        assert dist.rejected == [w] and dist.received == 0
        a.close()                                                # ... and it never completes
        for _ in range(200):
            if dist.leases:
                break
            time.sleep(0.01)
        assert [lw for lw, _ in dist.leases] == [w] and not dist.receiving   # lease restored
        assert worker.submit_workload("127.0.0.1", dist.port, w, np.ones(CHUNK_BYTES, np.uint8)) is True
        for _ in range(200):
            if dist.received == 1:
                break
            time.sleep(0.01)
        assert dist.received == 1 and dist.all_done()
