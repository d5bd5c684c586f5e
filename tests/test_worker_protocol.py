"""CPU integration tests of the worker's wire protocol against a Python restatement of the
reference's Distributer (tests/fake_distributer.py).  The compute function is injected (the product
default is the HIP path, which has no CPU fallback); here it is the CPU oracle or a pattern."""
import socket
import struct
import threading

import numpy as np
import pytest

from distributedmandelbrot_amd import worker
from fake_distributer import CHUNK_BYTES, FakeDistributer

QUIET = lambda *a: None  # noqa: E731


def pattern_compute(level, mrd, ir, ii):
    out = np.empty(CHUNK_BYTES, np.uint8)
    out[:] = (np.arange(CHUNK_BYTES, dtype=np.uint32) * 2654435761 >> 13).astype(np.uint8)
    out[:16] = np.frombuffer(struct.pack("<IIII", level, mrd, ir, ii), np.uint8)
    return out


def test_constants_match_reference():
    # WorkerCUDA.py:7-17, Program.cs:13, DataChunk.cs:20,27
    assert (worker.REQUEST_CODE, worker.RESPONSE_CODE) == (0x00, 0x01)
    assert (worker.WORKLOAD_AVAILABLE_CODE, worker.WORKLOAD_NOT_AVAILABLE_CODE) == (0x10, 0x11)
    assert (worker.WORKLOAD_ACCEPT_CODE, worker.WORKLOAD_REJECT_CODE) == (0x20, 0x21)
    assert worker.DEFAULT_DISTRIBUTER_PORT == 59010 and worker.CHUNK_BYTES == 16777216
    assert (worker.MIN_AXIS, worker.MAX_AXIS) == (-2, 2)


def test_single_tile_roundtrip_with_oracle_compute(oracle, golden):
    """BASELINE cfg1 at the wire level: DataChunk (level 1, 0, 0), mrd 256, one worker, through
    the Distributer protocol; the bytes that arrive equal the tile the reference's own process_workload
    produced (golden SHA-256)."""
    import hashlib

    def oracle_compute(level, mrd, ir, ii):
        return oracle.datachunk(level, mrd, ir, ii, want_counts=False)[1].ravel()

    with FakeDistributer([(1, 256)]) as srv:
        assert worker.do_workload_single("127.0.0.1", srv.port, compute=oracle_compute, log=QUIET) is True
        assert worker.do_workload_single("127.0.0.1", srv.port, compute=oracle_compute, log=QUIET) is False
        (w, data), = srv.completed.items()
        assert w == (1, 256, 0, 0)
        assert hashlib.sha256(data.tobytes()).hexdigest() == str(golden["full/1_256_0_0/bytes_sha256"])
        assert not srv.leases


def test_scan_order_and_all_tiles_completed():
    # level 2 -> 4 tiles handed out indexReal-major, indexImag-minor (Distributer.cs:338-340)
    with FakeDistributer([(2, 16), (1, 8)]) as srv:
        order = []

        def compute(level, mrd, ir, ii):
            order.append((level, mrd, ir, ii))
            return pattern_compute(level, mrd, ir, ii)

        n = 0
        while worker.do_workload_single("127.0.0.1", srv.port, compute=compute, log=QUIET):
            n += 1
        assert n == 5
        assert order == [(2, 16, 0, 0), (2, 16, 0, 1), (2, 16, 1, 0), (2, 16, 1, 1), (1, 8, 0, 0)]
        for w, data in srv.completed.items():
            assert np.array_equal(data, pattern_compute(*w))


def test_rejected_result_returns_true_and_tile_is_reissued():
    with FakeDistributer([(1, 32)]) as srv:
        def slow_compute(level, mrd, ir, ii):
            srv.expire_all_leases()  # the lease times out while we compute (Distributer.cs:22,153-160)
            return pattern_compute(level, mrd, ir, ii)

        # reject -> the reference returns True and carries on (WorkerCUDA.py:161-163)
        assert worker.do_workload_single("127.0.0.1", srv.port, compute=slow_compute, log=QUIET) is True
        assert srv.rejected == [(1, 32, 0, 0)] and not srv.completed
        # the expired tile is offered again and now completes
        assert worker.do_workload_single("127.0.0.1", srv.port, compute=pattern_compute, log=QUIET) is True
        assert srv.wait_completed(1)
        assert list(srv.completed) == [(1, 32, 0, 0)]


def _one_shot_server(reply: bytes):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    s.listen(1)

    def run():
        c, _ = s.accept()
        c.recv(1)
        c.sendall(reply)
        c.close()
        s.close()

    threading.Thread(target=run, daemon=True).start()
    return s.getsockname()[1]


def test_unknown_opcode_raises_like_the_reference():
    port = _one_shot_server(bytes([0x7F]))
    with pytest.raises(Exception, match="Unknown response code to request: 127"):
        worker.do_workload_single("127.0.0.1", port, compute=pattern_compute, log=QUIET)


def test_short_header_is_an_error_not_garbage():
    port = _one_shot_server(bytes([0x10, 1, 0]))  # workload header cut short
    with pytest.raises(ConnectionError):
        worker.request_workload("127.0.0.1", port)


def test_payload_size_is_enforced():
    with pytest.raises(ValueError):
        worker.submit_workload("127.0.0.1", 1, (1, 1, 0, 0), np.zeros(100, np.uint8))


def test_farm_two_feeders_complete_every_tile_once():
    with FakeDistributer([(3, 16)]) as srv:
        seen = []
        lock = threading.Lock()

        def make_compute(dev):
            def compute(level, mrd, ir, ii):
                with lock:
                    seen.append((dev, (level, mrd, ir, ii)))
                return pattern_compute(level, mrd, ir, ii)
            return compute

        done = worker.run_farm("127.0.0.1", srv.port, devices=[0, 1], make_compute=make_compute, log=QUIET)
        assert srv.wait_completed(9)
        assert sum(done) == 9 and len(srv.completed) == 9
        assert sorted(w for _, w in seen) == sorted(srv.completed)
        assert len({w for _, w in seen}) == 9  # no tile computed twice
        for w, data in srv.completed.items():
            assert np.array_equal(data, pattern_compute(*w))


def test_reference_faithful_single_receive_mode_documents_the_defect():
    # Distributer.cs:416 reads the 16 MiB with ONE Receive; our worker still sends all of it.
    with FakeDistributer([(1, 8)], faithful_single_receive=True) as srv:
        assert worker.do_workload_single("127.0.0.1", srv.port, compute=pattern_compute, log=QUIET)
        assert srv.wait_completed(1)
        (w, data), = srv.completed.items()
        got = int([l for l in srv.log if l.startswith("single receive")][0].split()[3])
        assert 0 < got <= CHUNK_BYTES
        assert np.array_equal(data[:got], pattern_compute(*w)[:got])


class _FakeDevice:
    """Stands in for MandelbrotDevice in run_pipelined (CPU tests): two slots, pattern bytes as 'compute'."""

    def __init__(self):
        from distributedmandelbrot_amd.device import TileStats
        self._stats = TileStats
        self.slots = [None, None]
        self.submitted = []
        self.max_inflight = 0

    def pinned_empty(self, shape, dtype):
        return np.empty(shape, dtype)

    def submit_datachunk(self, slot, level, mrd, ir, ii, out_bytes, lazy_uniform=False):
        assert self.slots[slot] is None, "slot reused before wait"
        self.lazy = getattr(self, "lazy", 0) + int(lazy_uniform)
        self.slots[slot] = ((level, mrd, ir, ii), out_bytes)
        self.submitted.append((level, mrd, ir, ii))
        self.max_inflight = max(self.max_inflight, sum(x is not None for x in self.slots))

    def wait(self, slot):
        w, buf = self.slots[slot]
        self.slots[slot] = None
        if self.uniform_value is not None:     # a uniform tile: like MBK_LAZY_UNIFORM, the buffer is NOT written
            return self._stats(0.1, 0.0, 1000, 0, self.uniform_value == 0, self.uniform_value == 1, 1)
        buf[:] = pattern_compute(*w)
        return self._stats(0.1, 0.1, 1000, 0, False, False, 5)

    uniform_value = None

    def close(self):
        pass


def test_pipelined_worker_same_wire_every_tile_once():
    """run_pipelined: lease / compute / send overlapped with two tiles in flight on the device; per tile
    the wire carries the reference's two exchanges, and every tile is completed exactly once."""
    with FakeDistributer([(3, 16), (1, 8)]) as srv:
        dev = _FakeDevice()
        n = worker.run_pipelined("127.0.0.1", srv.port, device=dev, log=QUIET, senders=2)
        assert srv.wait_completed(10)
        assert n == 10 and len(srv.completed) == 10 and not srv.rejected
        assert sorted(dev.submitted) == sorted(srv.completed) and len(set(dev.submitted)) == 10
        assert dev.max_inflight == 2                      # both device slots were really used
        for w, data in srv.completed.items():
            assert np.array_equal(data, pattern_compute(*w))
        assert not [l for l in srv.log if "error" in l or "unknown" in l], srv.log


def test_pipelined_worker_respects_max_tiles_and_threaded_server():
    from distributedmandelbrot_amd.server import Distributer
    with Distributer([(4, 16)]) as dist:
        dev = _FakeDevice()
        n = worker.run_pipelined("127.0.0.1", dist.port, device=dev, log=QUIET, senders=3, max_tiles=7)

        def settled(k):        # the client returns when its last byte is queued; the server may lag
            import time
            for _ in range(500):
                if dist.received == k:
                    return True
                time.sleep(0.01)
            return False
        assert n == 7 and settled(7)
        n2 = worker.run_pipelined("127.0.0.1", dist.port, device=_FakeDevice(), log=QUIET)
        assert n2 == 9 and settled(16) and dist.all_done()


def test_reset_mid_payload_is_reported_and_counted():
    """A server that stops reading after one Receive (the reference defect, Distributer.cs:416) resets the
    still-sending client: the worker reports SUBMIT_RESET with the bytes it got out, counts it, logs a
    warning -- and, like the reference worker, carries on."""
    before = dict(worker.stats)
    with FakeDistributer([(1, 8)], faithful_single_receive=True) as srv:
        w = worker.request_workload("127.0.0.1", srv.port)
        status, sent = worker.submit_workload_ex("127.0.0.1", srv.port, w, pattern_compute(*w))
        assert srv.wait_completed(1)
    assert status in (worker.SUBMIT_RESET, worker.SUBMIT_ACCEPTED)
    if status == worker.SUBMIT_RESET:
        assert 0 < sent < CHUNK_BYTES and worker.stats["resets"] == before["resets"] + 1
        lines = []
        worker._log_submit(status, sent, lambda *a: lines.append(" ".join(str(x) for x in a)))
        assert any("connection reset by the server after" in l for l in lines)
    else:
        assert sent == CHUNK_BYTES and worker.stats["accepted"] == before["accepted"] + 1


def test_main_honours_an_explicit_single_device(monkeypatch):
    """`worker ADDR PORT 3` must run on GPU 3 (round 1 sent every single-device invocation to GPU 0)."""
    seen = {}
    monkeypatch.setattr(worker, "run_farm", lambda addr, port, devices, **kw: seen.update(addr=addr, port=port, devices=devices) or [0])
    worker.main(["10.0.0.1", "59010", "3"])
    assert seen == {"addr": "10.0.0.1", "port": 59010, "devices": [3]}
    worker.main(["h", "1", "0,2,5"])
    assert seen["devices"] == [0, 2, 5]


def test_pipelined_worker_sends_constant_payload_for_uniform_tiles():
    """Tiles the device reports as all-0 / all-1 are not copied off the GPU (MBK_LAZY_UNIFORM); the wire still
    carries 16 777 216 bytes of that constant."""
    for value in (0, 1):
        with FakeDistributer([(2, 16)]) as srv:
            dev = _FakeDevice()
            dev.uniform_value = value
            n = worker.run_pipelined("127.0.0.1", srv.port, device=dev, log=QUIET, senders=2)
            assert n == 4 and srv.wait_completed(4) and dev.lazy == 4
            for data in srv.completed.values():
                assert data.size == CHUNK_BYTES and (data == value).all()
