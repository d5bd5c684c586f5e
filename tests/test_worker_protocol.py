"""CPU integration tests of the worker's wire protocol against a Python restatement of the
reference's Distributer (tests/fake_distributer.py).  The compute function is injected (the product
default is the HIP path, which has no CPU fallback); here it is the CPU oracle or a pattern."""
import socket
import struct
import threading
import time

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


# ------------------------------------------------------------------------------------------------------
# Round 3: the same loop in native code (mbk_feeder_run in libmbk_hip.so; mbk_worker_run binds it to a GPU).
# The protocol engine is driven here through a backend made of ctypes callbacks: the compute is a pattern
# (or the oracle) -- the product binding, mbk_worker_run, needs a GPU and is covered by the -m gpu tests.
# ------------------------------------------------------------------------------------------------------

class _NativeBackend:
    """mbk_feeder_ops over Python callbacks: two slots, pattern compute, optional uniform tiles."""

    def __init__(self, compute=pattern_compute, uniform_value=None, fail_submit_at=None):
        import ctypes as C
        from distributedmandelbrot_amd import _lib as L
        self.C, self.L = C, L
        self.compute, self.uniform_value, self.fail_submit_at = compute, uniform_value, fail_submit_at
        self.pending = [None, None]
        self.submitted, self.tiles, self.max_inflight, self.slots = [], [], 0, []
        self._bufs = {}
        self.ops = L.mbk_feeder_ops(None, L.FEEDER_SUBMIT(self._submit), L.FEEDER_WAIT(self._wait),
                                    L.FEEDER_ALLOC(self._alloc), L.FEEDER_RELEASE(self._release),
                                    L.FEEDER_ON_TILE(self._on_tile))

    def _alloc(self, user, n):
        buf = (self.C.c_uint8 * n)()
        addr = self.C.addressof(buf)
        self._bufs[addr] = buf
        return addr

    def _release(self, user, ptr):
        self._bufs.pop(ptr, None)

    def _submit(self, user, slot, level, mrd, ir, ii, h_bytes):
        if self.fail_submit_at is not None and len(self.submitted) == self.fail_submit_at:
            return 1
        assert self.pending[slot] is None, "slot reused before wait"
        self.pending[slot] = ((level, mrd, ir, ii), h_bytes)
        self.submitted.append((level, mrd, ir, ii))
        self.slots.append(slot)
        self.max_inflight = max(self.max_inflight, sum(p is not None for p in self.pending))
        return 0

    def _wait(self, user, slot, stats):
        w, ptr = self.pending[slot]
        self.pending[slot] = None
        st = stats.contents
        st.kernel_ms, st.d2h_ms, st.pixel_iterations, st.never_pixels = 0.5, 0.1, 1000, 0
        st.all_bytes_zero = st.all_bytes_one = 0
        st.rle_runs = 0
        if self.uniform_value is None:
            np.ctypeslib.as_array((self.C.c_uint8 * CHUNK_BYTES).from_address(ptr))[:] = self.compute(*w)
        else:   # MBK_LAZY_UNIFORM: the buffer is left untouched, the stats carry the constant
            st.all_bytes_zero, st.all_bytes_one = int(self.uniform_value == 0), int(self.uniform_value == 1)
        return 0

    def _on_tile(self, user, w, stats, status):
        self.tiles.append((tuple(w.contents), status))

    def run(self, port, max_tiles=0, senders=2, addr="127.0.0.1"):
        lib = self.L.load()
        rep = self.L.mbk_worker_report()
        rc = lib.mbk_feeder_run(self.C.byref(self.ops), addr.encode(), port, max_tiles, senders, self.C.byref(rep))
        return rc, rep, (lib.mbk_last_error(None) or b"").decode()


def test_native_feeder_same_wire_every_tile_once():
    """mbk_feeder_run against the restated Distributer: per tile the reference's two exchanges
    (WorkerCUDA.py:115-134,148-172), every tile completed exactly once with the backend's bytes, both slots used,
    and the loop ends on 0x11 like the reference worker (WorkerCUDA.py:127-129)."""
    with FakeDistributer([(3, 16), (1, 8)]) as srv:
        be = _NativeBackend()
        rc, rep, err = be.run(srv.port, senders=3)
        assert rc == 0, err
        assert srv.wait_completed(10)
        assert (rep.leased, rep.accepted, rep.rejected, rep.resets, rep.uniform_tiles) == (10, 10, 0, 0, 0)
        assert rep.pixel_iterations == 10 * 1000 and rep.seconds > 0
        assert sorted(be.submitted) == sorted(srv.completed) and len(set(be.submitted)) == 10
        assert be.max_inflight == 2 and be.slots[:4] == [0, 1, 0, 1]
        assert sorted(be.tiles) == sorted((w, 1) for w in srv.completed)
        for w, data in srv.completed.items():
            assert np.array_equal(data, pattern_compute(*w))
        assert not [l for l in srv.log if "error" in l or "unknown" in l], srv.log
        assert not be._bufs                                  # every result buffer was released


def test_native_feeder_max_tiles_threaded_server_and_oracle_bytes(oracle, golden):
    import hashlib
    from distributedmandelbrot_amd.server import Distributer

    def settled(dist, k):
        import time
        for _ in range(1000):
            if dist.received == k:
                return True
            time.sleep(0.01)
        return False

    with Distributer([(4, 16)]) as dist:
        rc, rep, err = _NativeBackend().run(dist.port, max_tiles=7, senders=4)
        assert rc == 0 and rep.leased == 7 and rep.accepted == 7 and settled(dist, 7), err
        rc, rep, err = _NativeBackend().run(dist.port)
        assert rc == 0 and rep.leased == 9 and settled(dist, 16) and dist.all_done(), err
    # one real tile through the native loop: the bytes that arrive are the reference's (golden SHA-256)
    with FakeDistributer([(1, 256)]) as srv:
        be = _NativeBackend(compute=lambda l, m, ir, ii: oracle.datachunk(l, m, ir, ii, want_counts=False)[1].ravel())
        rc, rep, err = be.run(srv.port)
        assert rc == 0 and rep.accepted == 1 and srv.wait_completed(1), err
        assert hashlib.sha256(srv.completed[(1, 256, 0, 0)].tobytes()).hexdigest() == str(golden["full/1_256_0_0/bytes_sha256"])


def test_native_feeder_uniform_tiles_rejects_and_errors():
    # uniform tiles: the wire still carries 16 777 216 bytes of the constant, from a shared buffer
    for value in (0, 1):
        with FakeDistributer([(2, 16)]) as srv:
            be = _NativeBackend(uniform_value=value)
            rc, rep, err = be.run(srv.port)
            assert rc == 0 and rep.uniform_tiles == 4 and srv.wait_completed(4), err
            for data in srv.completed.values():
                assert data.size == CHUNK_BYTES and (data == value).all()
    # a result whose lease expired meanwhile is rejected (0x21): dropped, the loop carries on (WorkerCUDA.py:161-163)
    with FakeDistributer([(1, 8)]) as srv:
        be = _NativeBackend()
        orig = be._wait

        def wait_and_expire(user, slot, stats):
            srv.expire_all_leases()
            return orig(user, slot, stats)
        be.ops.wait = be.L.FEEDER_WAIT(wait_and_expire)
        rc, rep, err = be.run(srv.port, max_tiles=1)
        assert rc == 0 and rep.leased == 1 and rep.rejected == 1 and rep.accepted == 0, err
        assert be.tiles == [((1, 8, 0, 0), 0)] and srv.rejected == [(1, 8, 0, 0)]
    # nobody listening: MBK_ERR_NET with the reason, not a hang and not a silent zero
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        dead_port = s_.getsockname()[1]
    rc, rep, err = _NativeBackend().run(dead_port)
    assert rc == 5 and "connect" in err and rep.leased == 0
    # a server that speaks out of protocol (WorkerCUDA.py:131-132 raises on an unknown reply)
    srv_sock = socket.socket()
    srv_sock.bind(("127.0.0.1", 0))
    srv_sock.listen(1)

    def bad_server():
        c, _ = srv_sock.accept()
        c.recv(1)
        c.sendall(bytes([0x77]))
        c.close()
    th = threading.Thread(target=bad_server, daemon=True)
    th.start()
    rc, rep, err = _NativeBackend().run(srv_sock.getsockname()[1])
    th.join(timeout=5)
    srv_sock.close()
    assert rc == 5 and "Unknown response code to request: 119" in err
    # a backend failure while tiles are in flight: the tiles already computed still arrive, the status is the backend's
    with FakeDistributer([(2, 16)]) as srv:
        be = _NativeBackend(fail_submit_at=2)
        rc, rep, err = be.run(srv.port)
        assert rc == 1 and "backend submit failed" in err and rep.accepted == 2 and srv.wait_completed(2)


def test_native_feeder_against_the_single_receive_server():
    """The reference server reads the payload with ONE Receive and closes (Distributer.cs:416-423): the still-sending
    client may be reset although the tile is complete on the server.  Counted as a reset, not an error."""
    with FakeDistributer([(1, 8)], faithful_single_receive=True) as srv:
        rc, rep, err = _NativeBackend().run(srv.port)
        assert rc == 0 and rep.leased == 1 and rep.accepted + rep.resets == 1 and srv.wait_completed(1), err


def test_run_farm_refuses_an_empty_device_list(monkeypatch):
    """`worker ADDR PORT` on a box without a visible GPU must raise, not print 'tiles: 0' and exit 0."""
    from distributedmandelbrot_amd import device
    monkeypatch.setattr(device, "device_count", lambda: 0)
    with pytest.raises(RuntimeError, match="no gfx950 GPU visible"):
        worker.main(["127.0.0.1", "1"])
    with pytest.raises(RuntimeError, match="no gfx950 GPU visible"):
        worker.run_farm("127.0.0.1", 1, devices=[])


def test_pipelined_worker_drains_inflight_tiles_when_the_lease_connection_fails():
    """an exception from request_workload must not drop the tiles already on the device: they are waited
    for and sent, the device's slots end up free, and the error is re-raised afterwards."""
    calls = {"n": 0}
    real = worker.request_workload

    def flaky(addr, port, timeout=None):
        calls["n"] += 1
        if calls["n"] == 3:
            raise ConnectionRefusedError("server restarting")
        return real(addr, port, timeout)

    with FakeDistributer([(2, 16)]) as srv:
        dev = _FakeDevice()
        worker.request_workload = flaky
        try:
            with pytest.raises(ConnectionRefusedError):
                worker.run_pipelined("127.0.0.1", srv.port, device=dev, log=QUIET, senders=2)
        finally:
            worker.request_workload = real
        assert srv.wait_completed(2) and len(dev.submitted) == 2
        assert all(p is None for p in dev.slots)


# ------------------------------------------------------------------------------------------------------
# Round 4: the return path against a server shaped like the REAL one: one accept thread,
# backlog 16, 100 ms receive timeouts, >= 10 ms per payload, resets when it is overrun.
# ------------------------------------------------------------------------------------------------------

def _net(name, value=None):
    import ctypes as C
    from distributedmandelbrot_amd import _lib as L
    lib = L.load()
    if value is None:
        v = C.c_uint32(0)
        assert lib.mbk_net_get_option(L.NET_OPTIONS[name], C.byref(v)) == 0
        return v.value
    assert lib.mbk_net_set_option(L.NET_OPTIONS[name], value) == 0


def test_farm_of_native_feeders_against_a_reference_shaped_server():
    """8 feeders x 4 senders (+ 8 lease connections) = up to 40 threads that want a connection, against ONE serial accept
    loop with backlog 16 that resets every 7th connection: the process-wide gate keeps at most 8 connections open, a
    reset exchange is repeated, and every tile is accepted exactly once with no lease left behind."""
    _net("max_connections", 8)
    _net("backoff_ms", 5)
    try:
        with FakeDistributer([(6, 16)], receive_timeout=0.1, payload_seconds=0.010, rst_every=7) as srv:
            backends = [_NativeBackend() for _ in range(8)]
            results = [None] * 8

            def feeder(k):
                results[k] = backends[k].run(srv.port, senders=4)
            threads = [threading.Thread(target=feeder, args=(k,)) for k in range(8)]
            for t in threads:
                t.start()
            for t in threads:
                t.join(timeout=120)
            assert all(r is not None and r[0] == 0 for r in results), [r and (r[0], r[2]) for r in results]
            assert srv.wait_completed(36)
            reps = [r[1] for r in results]
            assert sum(r.leased for r in reps) == 36 and sum(r.accepted for r in reps) == 36
            assert sum(r.rejected for r in reps) == 0 and sum(r.resets for r in reps) == 0
            assert sorted(w for be in backends for w in be.submitted) == sorted(srv.completed) and len(srv.completed) == 36
            assert srv.duplicate_completions == 0 and not srv.leases and not srv.rejected      # no lease lost, none left
            assert srv.resets_sent >= 5 and sum(r.net_retries for r in reps) == srv.resets_sent  # each reset cost one repeat
            assert 1 <= _net("peak_connections") <= 8
            for w, data in srv.completed.items():
                assert np.array_equal(data, pattern_compute(*w))
    finally:
        _net("backoff_ms", 50)


def test_native_feeder_times_out_on_a_server_that_accepts_and_never_answers():
    """a server that accepts but never answers must end the call with MBK_ERR_NET, not hang it.
    ADVICE r4: ... and the lease request is NOT repeated after such a timeout -- the request byte was sent, the server
    registers a lease as it answers, so a reply that was sent but not read in time would orphan a tile and the retry would
    take a second one (net_retries == 0).  A peer that closes before its reply never reached the hand-out code: repeated."""
    srv_sock = socket.socket()
    srv_sock.bind(("127.0.0.1", 0))
    srv_sock.listen(4)
    _net("io_timeout_ms", 200)
    _net("retries", 1)
    _net("backoff_ms", 5)
    try:
        t0 = time.monotonic()
        rc, rep, err = _NativeBackend().run(srv_sock.getsockname()[1])
        assert rc == 5 and rep.leased == 0 and rep.net_retries == 0 and time.monotonic() - t0 < 5.0
        assert "workload request" in err and "Success" not in err, err
    finally:
        _net("io_timeout_ms", 30000)
        _net("retries", 6)
        _net("backoff_ms", 50)
        srv_sock.close()
    # a peer that closes before its reply is named as such (round 3 printed 'workload request: Success')
    srv_sock = socket.socket()
    srv_sock.bind(("127.0.0.1", 0))
    srv_sock.listen(4)

    def closer():
        for _ in range(2):
            c, _ = srv_sock.accept()
            c.recv(1)
            c.close()
    th = threading.Thread(target=closer, daemon=True)
    th.start()
    _net("retries", 1)
    _net("backoff_ms", 5)
    try:
        rc, rep, err = _NativeBackend().run(srv_sock.getsockname()[1])
        assert rc == 5 and "connection closed by peer" in err, err
    finally:
        _net("retries", 6)
        _net("backoff_ms", 50)
        th.join(timeout=5)
        srv_sock.close()


def test_native_feeder_stop_flag_drains_and_ends():
    with FakeDistributer([(3, 16)]) as srv:
        be = _NativeBackend()
        orig = be._wait
        seen = {"n": 0}

        def wait_then_stop(user, slot, stats):
            seen["n"] += 1
            if seen["n"] == 2:
                _net("stop", 1)
            return orig(user, slot, stats)
        be.ops.wait = be.L.FEEDER_WAIT(wait_then_stop)
        try:
            rc, rep, err = be.run(srv.port)
        finally:
            _net("stop", 0)
        assert rc == 0 and 2 <= rep.leased <= 4 and rep.accepted == rep.leased and srv.wait_completed(rep.leased), err
        assert not srv.leases          # everything that was leased came back


def test_python_worker_retries_a_reset_connection_and_gates_its_connections():
    """The Python loops go through the same policy: connection gate, timeouts, retry with backoff until the server has
    answered; run_native(max_tiles=0) leases nothing."""
    old = (worker.NET.max_connections, worker.NET.backoff)
    worker.set_network_options(max_connections=2, backoff=0.005, native=False)
    try:
        with FakeDistributer([(2, 16)], receive_timeout=0.1, rst_every=3) as srv:
            before = worker.NET.retried
            done = worker.run_farm("127.0.0.1", srv.port, devices=[0, 1, 2], make_compute=lambda d: pattern_compute, log=QUIET)
            assert sum(done) == 4 and srv.wait_completed(4) and not srv.leases and not srv.rejected
            assert srv.resets_sent >= 2 and worker.NET.retried - before == srv.resets_sent
    finally:
        worker.set_network_options(max_connections=old[0], backoff=old[1], native=False)
    assert worker.run_native("127.0.0.1", 1, max_tiles=0, device=object()) == 0
