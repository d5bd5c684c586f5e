"""The tile worker: a drop-in for DistributedMandelbrotWorkerCUDA.py ("WorkerCUDA.py") that speaks
the reference's worker<->Distributer TCP protocol UNCHANGED and computes tiles on MI355X GPUs.

Same public names, argument meaning and return/raise behaviour as the reference script:

    receive_workload(sock)                 WorkerCUDA.py:102-109
    process_workload(level, mrd, ir, ii)   WorkerCUDA.py:70-100   -> np.ndarray uint8[16777216]
    do_workload_single(addr, port) -> bool WorkerCUDA.py:111-176
    main()                                 WorkerCUDA.py:178-184

Wire protocol (Distributer.cs:30-45,358-458; DistributerWorkload.cs:53-100; all little-endian u32):
    request : C->S 0x00 ; S->C 0x10 + level,mrd,indexReal,indexImag (4 x u32) | 0x11 (no work)
    response: C->S 0x01 + pack("IIII", level,mrd,indexReal,indexImag) ; S->C 0x20 | 0x21 ;
              on 0x20 C->S exactly 16 777 216 raw bytes (row = imaginary index, col = real index).

Against the real server (round 4): the reference Distributer accepts on ONE thread with a listen backlog of 16
(Distributer.cs:16,221,226-297) and 100 ms receive timeouts (:17,196-202); the reference worker held one connection
at a time.  A farm must not bury it under 8 x (senders + 1) concurrent connects, and a computed tile must not be lost
because one connect was refused: every connection here goes through a process-wide gate (`NET.max_connections`,
default 8), has connect / IO timeouts, and both exchanges are retried with exponential backoff on transient failures
until the server has answered (`set_network_options`; the native loop has the same knobs: mbk_net_set_option).

Differences from the reference worker, all wire-compatible:
  * `sendall` / receive-exactly instead of bare `send` / `recv(4)` (WorkerCUDA.py:104-107,168 may
    transfer short; the bytes on the wire are identical when nothing is cut short);
  * `run_farm`: one feeder thread per GPU, each an ordinary protocol client (the Distributer accepts
    any number of clients, Distributer.cs:226-297) -- the per-GPU work queue IS the Distributer's
    lease table, so there is no RCCL and no GPU<->GPU traffic;
  * `run_pipelined` (what run_farm runs per GPU): the same two exchanges per tile, but overlapped --
    while tile n is on the GPU, tile n+1 is being leased and tile n-1 is being sent by a sender thread
    (up to MBK_SLOTS = 4 tiles in flight on the device: mbk_datachunk_submit / mbk_wait); the reference is strictly
    lease -> compute -> send (WorkerCUDA.py:111-176);
  * there is no CPU fallback: without libmbk_hip.so and a gfx950 GPU, process_workload raises.
"""
from __future__ import annotations

import socket
import struct
import sys
import threading
import time
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

MIN_AXIS = -2  # WorkerCUDA.py:7
MAX_AXIS = 2   # WorkerCUDA.py:8

REQUEST_CODE = 0x00   # WorkerCUDA.py:10, Distributer.cs:30
RESPONSE_CODE = 0x01  # WorkerCUDA.py:11, Distributer.cs:31

WORKLOAD_AVAILABLE_CODE = 0x10      # WorkerCUDA.py:13, Distributer.cs:35
WORKLOAD_NOT_AVAILABLE_CODE = 0x11  # WorkerCUDA.py:14, Distributer.cs:38

WORKLOAD_ACCEPT_CODE = 0x20  # WorkerCUDA.py:16, Distributer.cs:42
WORKLOAD_REJECT_CODE = 0x21  # WorkerCUDA.py:17, Distributer.cs:45

DEFAULT_DISTRIBUTER_PORT = 59010  # Program.cs:13
CHUNK_DEFINITION = 4096           # WorkerCUDA.py:80, DataChunk.cs:20
CHUNK_BYTES = CHUNK_DEFINITION * CHUNK_DEFINITION  # DataChunk.cs:27, Distributer.cs:415-416

Workload = Tuple[int, int, int, int]  # (level, mrd, indexReal, indexImag)
ComputeFn = Callable[[int, int, int, int], np.ndarray]

_default_device = None
_default_pinned = None
_default_lock = threading.Lock()


def _get_default_device():
    global _default_device, _default_pinned
    with _default_lock:
        if _default_device is None:
            from .device import MandelbrotDevice  # raises if the HIP library / GPU is missing
            _default_device = MandelbrotDevice(0)
            _default_pinned = _default_device.pinned_empty((CHUNK_BYTES,), np.uint8)
        return _default_device


def process_workload(level: int, mrd: int, index_real: int, index_imag: int) -> np.ndarray:
    """WorkerCUDA.py:70-100 on the default GPU: the tile's 16 777 216 quantised bytes (a fresh array,
    like the reference's)."""
    out, _, _ = _get_default_device().datachunk(level, mrd, index_real, index_imag)
    return out


def _process_workload_pinned(level: int, mrd: int, index_real: int, index_imag: int) -> np.ndarray:
    """Same tile, but DMA'd straight into one reused pinned buffer (valid until the next call): what
    do_workload_single uses, since it sends the bytes before asking for the next tile."""
    dev = _get_default_device()
    out, _, st = dev.datachunk(level, mrd, index_real, index_imag, out_bytes=_default_pinned)
    _last_stats["default"] = st
    return out


_last_stats = {}  # device label -> TileStats of the last tile (the reference has no metrics at all)


def describe_stats(st) -> str:
    """One log line for a finished tile: kernel time, throughput, and what the server will do with it."""
    kind = "Never" if st.all_bytes_zero else "Immediate" if st.all_bytes_one else \
        ("RLE" if 1 + 5 * st.rle_runs < 1 + CHUNK_BYTES else "Raw")
    rate = st.pixel_iterations / st.kernel_ms / 1e6 if st.kernel_ms > 0 else 0.0
    # pixel_iterations are the REFERENCE's iterations for this output (count, or mrd-1 for a pixel of the set); with
    # the library's cycle test on, fewer steps are executed, so the rate is labelled reference-equivalent
    return (f"kernel {st.kernel_ms:.3f} ms, D2H {st.d2h_ms:.3f} ms, {st.pixel_iterations / 1e9:.2f} G pixel-iterations "
            f"({rate:.0f} G/s reference-equivalent), {st.never_pixels} in-set pixels, stored as {kind}")


class _Net:
    """Process-wide network behaviour (the Python loops; `set_network_options` mirrors it into the native loop)."""
    max_connections = 8          # < the reference's listenBacklog of 16 (Distributer.cs:16)
    connect_timeout = 10.0       # seconds
    io_timeout = 30.0            # per send / recv call without progress
    retries = 6                  # attempts after the first, per exchange
    backoff = 0.05               # first pause; doubles per attempt up to 2 s, plus jitter
    retried = 0                  # exchanges repeated after a transient failure (diagnostic)
    lease_timeouts = 0           # lease requests whose reply timed out after the request was sent (never retried)
    _gate = threading.BoundedSemaphore(8)


NET = _Net


def set_network_options(max_connections: Optional[int] = None, connect_timeout: Optional[float] = None,
                        io_timeout: Optional[float] = None, retries: Optional[int] = None,
                        backoff: Optional[float] = None, native: bool = True) -> None:
    """Tune the connection gate, the timeouts and the retry policy of both worker loops (Python and, when the library
    is built, native).  Call before starting a farm: resizing the gate while connections are open is not supported."""
    if max_connections is not None:
        if not 1 <= max_connections <= 64:
            raise ValueError("max_connections must be in 1..64")
        NET.max_connections = int(max_connections)
        NET._gate = threading.BoundedSemaphore(NET.max_connections)
    if connect_timeout is not None:
        NET.connect_timeout = float(connect_timeout)
    if io_timeout is not None:
        NET.io_timeout = float(io_timeout)
    if retries is not None:
        NET.retries = int(retries)
    if backoff is not None:
        NET.backoff = float(backoff)
    if native:
        try:
            from . import _lib as L
            lib = L.load()
        except (ImportError, OSError):
            return
        for name, value in (("max_connections", NET.max_connections), ("connect_timeout_ms", int(NET.connect_timeout * 1e3)),
                            ("io_timeout_ms", int(NET.io_timeout * 1e3)), ("retries", NET.retries),
                            ("backoff_ms", max(1, int(NET.backoff * 1e3)))):
            if lib.mbk_net_set_option(L.NET_OPTIONS[name], value) != L.MBK_OK:
                raise ValueError(f"native loop rejected {name} = {value}")


_stop_event = threading.Event()   # the Python loops' counterpart of MBK_NET_STOP


def request_stop(stop: bool = True) -> None:
    """Ask every running loop -- native (mbk_worker_run is one blocking C call per GPU: MBK_NET_STOP) and Python
    (run_pipelined and the serial loop of run_farm check a module-level event before every lease) -- to stop leasing,
    return the tiles it holds and end.  request_stop(False) re-arms; run_farm and main do so when they start, so a stop
    from an earlier run in the same process does not end the next one at once (the flag is process-wide)."""
    if stop:
        _stop_event.set()
    else:
        _stop_event.clear()
    try:
        from . import _lib as L
        lib = L.load()
    except (ImportError, OSError):
        return          # no native library: only the Python loops exist
    lib.mbk_net_set_option(L.NET_OPTIONS["stop"], 1 if stop else 0)


# "not now" rather than "never": the server's backlog was full (refused / reset), it was busy past a timeout, or it
# closed before answering.  Anything else (e.g. an unknown reply code) is an error at once, as in the reference.
_TRANSIENT = (ConnectionRefusedError, ConnectionResetError, ConnectionAbortedError, BrokenPipeError, socket.timeout,
              TimeoutError, InterruptedError)


class _PeerClosed(ConnectionError):
    pass


def _with_retries(exchange: Callable[[socket.socket], object], addr: str, port: int, timeout: Optional[float]):
    """Run `exchange(sock)` on a fresh connection, through the connection gate; repeat it with exponential backoff while
    it fails in a transient way BEFORE the server has answered (the exchange raises _Answered-wrapped errors itself
    once it must not be repeated)."""
    import random
    last: Optional[BaseException] = None
    for attempt in range(1 + max(0, NET.retries)):
        if attempt:
            pause = min(2.0, NET.backoff * (1 << min(attempt - 1, 6)))
            time.sleep(pause + random.random() * pause / 2)
            NET.retried += 1
        gate = NET._gate
        with gate:
            try:
                with socket.create_connection((addr, port), timeout=NET.connect_timeout if timeout is None else timeout) as sock:
                    sock.settimeout(NET.io_timeout if timeout is None else timeout)
                    return exchange(sock)
            except _NoRetry as e:
                raise e.inner from None
            except (_PeerClosed,) + _TRANSIENT as e:
                last = e
            except OSError as e:
                import errno
                if e.errno not in (errno.EHOSTUNREACH, errno.ENETUNREACH, errno.EAGAIN, errno.ETIMEDOUT):
                    raise
                last = e
    assert last is not None
    raise last


class _NoRetry(Exception):
    """Wraps a failure that happened after the server answered: the exchange must not be repeated."""

    def __init__(self, inner: BaseException):
        super().__init__(str(inner))
        self.inner = inner


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        part = sock.recv(n - len(buf))
        if not part:
            raise _PeerClosed(f"connection closed by peer after {len(buf)} of {n} bytes")
        buf += part
    return bytes(buf)


def receive_workload(sock: socket.socket) -> Workload:
    """WorkerCUDA.py:102-109: four little-endian u32 (DistributerWorkload.cs:53-77 sends them as four
    separate 4-byte sends)."""
    level = struct.unpack("<I", _recv_exact(sock, 4))[0]
    mrd = struct.unpack("<I", _recv_exact(sock, 4))[0]
    index_real = struct.unpack("<I", _recv_exact(sock, 4))[0]
    index_imag = struct.unpack("<I", _recv_exact(sock, 4))[0]
    return level, mrd, index_real, index_imag


def request_workload(addr: str, port: int, timeout: Optional[float] = None) -> Optional[Workload]:
    """First connection of WorkerCUDA.py:115-134.  None == 0x11 (no workload available).  A connect that is refused /
    reset / times out, or a server that closes (or resets) before its reply byte, is retried with backoff (NET.retries);
    a reply that does not arrive within the I/O timeout AFTER the request byte was sent is an error at once."""
    def exchange(sock: socket.socket):
        sock.sendall(struct.pack("B", REQUEST_CODE))
        try:
            response = _recv_exact(sock, 1)[0]
        except (socket.timeout, TimeoutError) as e:
            # The request byte is out and no reply came in time: the server may have sent 0x10 + a workload that we did
            # not read, and it registers the lease as it sends (Distributer.cs HandleWorkloadRequest) -- asking again
            # would take a second tile and orphan the first for its one-hour lease.  Not retried; an orderly
            # close or a reset before the reply byte still is (the hand-out code was never reached).
            NET.lease_timeouts += 1
            raise _NoRetry(e)
        try:
            if response == WORKLOAD_AVAILABLE_CODE:
                return receive_workload(sock)      # the lease exists on the server now: never ask again for this one
            if response == WORKLOAD_NOT_AVAILABLE_CODE:
                return None
            raise Exception("Unknown response code to request: " + str(response))  # WorkerCUDA.py:131-132
        except BaseException as e:
            raise _NoRetry(e)
    return _with_retries(exchange, addr, port, timeout)


SUBMIT_REJECTED, SUBMIT_ACCEPTED, SUBMIT_RESET = 0, 1, 2
stats = {"accepted": 0, "rejected": 0, "resets": 0}   # process-wide tile counters (the reference has none)
_stats_lock = threading.Lock()


def _count(key: str) -> None:
    with _stats_lock:
        stats[key] += 1


def submit_workload_ex(addr: str, port: int, workload: Workload, out: np.ndarray,
                       timeout: Optional[float] = None) -> Tuple[int, int]:
    """Second connection of WorkerCUDA.py:148-172.  Returns (status, payload bytes handed to the socket):
    SUBMIT_REJECTED (0x21), SUBMIT_ACCEPTED (0x20 and every byte sent) or SUBMIT_RESET (0x20, then the
    server reset the connection mid-payload)."""
    payload = memoryview(np.ascontiguousarray(out, dtype=np.uint8)).cast("B")
    if len(payload) != CHUNK_BYTES:
        raise ValueError(f"tile payload must be {CHUNK_BYTES} bytes, got {len(payload)}")
    def exchange(sock: socket.socket) -> Tuple[int, int]:
        sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        # one segment for opcode + header: the server reads them with separate 100 ms-timeout
        # receives (Distributer.cs:17,243-245,400), so do not dribble them
        sock.sendall(struct.pack("<BIIII", RESPONSE_CODE, *workload))
        response = _recv_exact(sock, 1)[0]
        # from here on the server has answered: nothing below may be repeated (on 0x20 it removed the lease,
        # Distributer.cs:404-423)
        try:
            if response == WORKLOAD_REJECT_CODE:
                _count("rejected")
                return SUBMIT_REJECTED, 0
            if response != WORKLOAD_ACCEPT_CODE:
                raise Exception("Unknown response code to request: " + str(response))  # WorkerCUDA.py:165-166
            sent = 0
            try:
                while sent < CHUNK_BYTES:  # exactly 16 777 216 raw bytes, no header (WorkerCUDA.py:168)
                    sent += sock.send(payload[sent:])
            except (ConnectionResetError, BrokenPipeError):
                # The reference server reads the payload with ONE Socket.Receive (Distributer.cs:416) and
                # then closes; with unread bytes in flight that close is a TCP reset.  By then it has
                # already marked the tile completed (:422-423), and the reference worker's single
                # `sock.send` never notices.  A server that died mid-transfer looks exactly the same from
                # here, so the event is reported to the caller (and counted) instead of being swallowed.
                _count("resets")
                return SUBMIT_RESET, sent
            _count("accepted")
            return SUBMIT_ACCEPTED, sent
        except BaseException as e:
            raise _NoRetry(e)
    return _with_retries(exchange, addr, port, timeout)


def submit_workload(addr: str, port: int, workload: Workload, out: np.ndarray,
                    timeout: Optional[float] = None) -> bool:
    """True == accepted (0x20), False == rejected (0x21).  A reset after 0x20 counts as accepted, as it
    does for the reference worker (see submit_workload_ex, which also says how much was sent)."""
    return submit_workload_ex(addr, port, workload, out, timeout)[0] != SUBMIT_REJECTED


def _log_submit(status: int, sent: int, log: Callable[..., None]) -> None:
    if status == SUBMIT_ACCEPTED:
        log("Response accepted")
        log("Sent response")
    elif status == SUBMIT_RESET:
        log("Response accepted")
        log(f"WARNING: connection reset by the server after {sent} of {CHUNK_BYTES} payload bytes. The reference "
            "Distributer reads the payload with a single Receive and then closes (Distributer.cs:416-423), which "
            "resets a still-sending client although the tile is already marked complete; a server that died "
            "looks the same -- then the tile stays leased until its 1 h lease expires (Distributer.cs:22).")
    else:
        log("Response rejected")


def do_workload_single(addr: str, port: int, compute: Optional[ComputeFn] = None,
                       log: Callable[..., None] = print) -> bool:
    """WorkerCUDA.py:111-176.  Returns False when the server has no workload (the reference then
    ends the program), True otherwise -- including when the result was rejected (:161-163)."""
    workload = request_workload(addr, port)
    if workload is None:
        log("No workload was available, ending program")
        return False
    log("Workload received:", workload)
    log("Starting calculation...")
    t0 = time.perf_counter()
    out = (compute or _process_workload_pinned)(*workload)
    log("Calculation complete (%.1f ms)" % ((time.perf_counter() - t0) * 1e3))
    if compute is None and "default" in _last_stats:
        log(describe_stats(_last_stats["default"]))
    status, sent = submit_workload_ex(addr, port, workload, out)
    _log_submit(status, sent, log)
    log("Process complete")
    return True


def run_pipelined(addr: str, port: int, device_index: int = 0, log: Callable[..., None] = print,
                  max_tiles: Optional[int] = None, senders: int = 2, device=None,
                  take: Optional[Callable[[], bool]] = None) -> int:
    """One GPU, the protocol of do_workload_single, three stages overlapped:

        lease tile n+1  |  GPU: tile n (slot n % SLOTS) -- D2H of earlier tiles overlaps it  |  sender thread(s): tile n-1

    Per tile the wire sees exactly the reference's two exchanges (WorkerCUDA.py:115-134 and :148-172);
    only their timing overlaps with other tiles', which the Distributer allows (any connection may
    return any leased tile).  The device is driven from this thread only (an mbk_ctx is not
    thread-safe); sender threads touch sockets and pinned host buffers.  `senders + SLOTS` pinned 16 MiB
    buffers circulate; a tile is submitted to the GPU only when a buffer is free, so a slow server
    back-pressures the lease rate instead of growing a queue.  Ends when the server answers 0x11 or after
    `max_tiles`; returns the number of tiles sent (accepted, incl. resets)."""
    import queue
    from .device import MandelbrotDevice
    own = device is None
    dev = device if device is not None else MandelbrotDevice(device_index)
    nslots = int(getattr(dev, "WORKER_DEPTH", getattr(dev, "SLOTS", 2)))   # MBK_WORKER_DEPTH = 3 of the MBK_SLOTS = 4
    nbuf = senders + nslots
    free: "queue.Queue" = queue.Queue()
    for _ in range(nbuf):
        free.put(dev.pinned_empty((CHUNK_BYTES,), np.uint8))
    outbox: "queue.Queue" = queue.Queue()
    errors: List[BaseException] = []
    sent_ok = [0]

    def sender() -> None:
        while True:
            item = outbox.get()
            if item is None:
                return
            workload, buf, st = item
            try:
                status, nsent = submit_workload_ex(addr, port, workload, payload_of(buf, st))
                log(f"{workload}: " + describe_stats(st))
                _log_submit(status, nsent, log)
                if status != SUBMIT_REJECTED:
                    with _stats_lock:
                        sent_ok[0] += 1
            except BaseException as e:
                errors.append(e)
            finally:
                free.put(buf)

    # uniform tiles (3 in 4 of a pyramid level) are never copied off the GPU: the stats say which constant the
    # tile is, and the 16 777 216 payload bytes the protocol wants come from one of two shared buffers
    constant = {0: None, 1: None}

    def payload_of(buf: np.ndarray, st) -> np.ndarray:
        value = 0 if st.all_bytes_zero else 1 if st.all_bytes_one else None
        if value is None:
            return buf
        if constant[value] is None:
            constant[value] = np.full(CHUNK_BYTES, value, np.uint8)
        return constant[value]

    threads = [threading.Thread(target=sender, daemon=True) for _ in range(max(1, senders))]
    for t in threads:
        t.start()
    inflight: List[Optional[tuple]] = [None] * nslots   # per device slot: (workload, buffer)
    leased = 0
    slot = 0
    try:
        more = True
        while more or any(inflight):
            if (more and not errors and not _stop_event.is_set() and (max_tiles is None or leased < max_tiles)
                    and (take is None or take())):
                try:
                    workload = request_workload(addr, port)
                except BaseException as e:   # e.g. connection refused while the server restarts: stop leasing, but
                    errors.append(e)         # finish (wait for + send) the tiles already leased; re-raised below
                    workload, more = None, False
                if workload is None and more:
                    log("No workload was available, ending program")
                    more = False
            else:
                workload, more = None, False
            # retire the tile that occupies the slot we are about to reuse (or drain at the end)
            if inflight[slot] is not None:
                w_old, b_old = inflight[slot]
                st = dev.wait(slot)
                inflight[slot] = None
                outbox.put((w_old, b_old, st))
            if workload is not None:
                log("Workload received:", workload)
                buf = free.get()
                dev.submit_datachunk(slot, *workload, buf, lazy_uniform=True)
                inflight[slot] = (workload, buf)
                leased += 1
            slot = (slot + 1) % nslots
    finally:
        for _ in threads:
            outbox.put(None)
        for t in threads:
            t.join()
        if own:
            dev.close()
    if errors:
        raise errors[0]
    return sent_ok[0]


def run_native(addr: str, port: int, device_index: int = 0, log: Callable[..., None] = print,
               max_tiles: Optional[int] = None, senders: int = 4, device=None) -> int:
    """run_pipelined's loop in native code: ONE call into libmbk_hip.so (mbk_worker_run, include/mbk.h) leases,
    computes and sends tiles until the server answers 0x11 or `max_tiles` were leased -- same wire traffic, no
    Python on the per-tile path (the GIL is released for the whole call; sender threads are C++ threads).
    Returns the number of tiles sent (accepted, incl. resets); the process-wide `stats` are updated."""
    import ctypes as C
    from . import _lib as L
    from .device import MandelbrotDevice, MbkError
    if max_tiles is not None and max_tiles <= 0:
        return 0        # (mbk_worker_run reads 0 as "no limit": only None may map to it)
    own = device is None
    dev = device if device is not None else MandelbrotDevice(device_index)
    rep = L.mbk_worker_report()
    try:
        st = dev._lib.mbk_worker_run(dev._h, addr.encode(), port, max_tiles or 0, max(1, senders), C.byref(rep))
        with _stats_lock:
            stats["accepted"] += rep.accepted
            stats["rejected"] += rep.rejected
            stats["resets"] += rep.resets
        rate = rep.leased / rep.seconds if rep.seconds > 0 else 0.0
        log(f"native feeder: {rep.leased} tiles leased, {rep.accepted} accepted, {rep.rejected} rejected, {rep.resets} "
            f"reset after accept, {rep.uniform_tiles} uniform (not copied off the GPU); {rep.seconds:.3f} s = {rate:.1f} "
            f"tiles/s; kernel time {rep.kernel_ms_sum:.1f} ms; {rep.pixel_iterations / 1e9:.2f} G pixel-iterations "
            f"(reference-equivalent); {rep.net_retries} exchange(s) repeated after a transient network failure")
        if st != L.MBK_OK:
            raise MbkError(st, (dev._lib.mbk_last_error(dev._h) or b"").decode())
        return int(rep.accepted + rep.resets)
    finally:
        if own:
            dev.close()


def run_farm(addr: str, port: int, devices: Optional[Sequence[int]] = None,
             make_compute: Optional[Callable[[int], ComputeFn]] = None,
             log: Callable[..., None] = print, max_tiles: Optional[int] = None, senders: int = 2,
             native: bool = False) -> List[int]:
    """One feeder thread per GPU until the Distributer answers 0x11.  Each feeder is `run_pipelined`
    (lease / compute / send overlapped, four tiles in flight on its GPU) or, with native=True, `run_native` (the
    same loop inside libmbk_hip.so; `max_tiles` is then split evenly over the feeders up front); with
    `make_compute(device_index)` -- a per-thread compute function, used by the CPU tests -- it is the serial
    do_workload_single loop.  Returns the number of tiles each feeder completed."""
    if devices is None:
        from .device import device_count
        devices = list(range(device_count()))
    if not devices:   # also an explicitly empty list: a farm node that silently does nothing is a misconfiguration
        raise RuntimeError("no gfx950 GPU visible and no CPU fallback exists")
    request_stop(False)   # a stop left over from an earlier run in this process must not end this one

    done = [0] * len(devices)
    errors: List[BaseException] = []
    budget = [max_tiles]
    budget_lock = threading.Lock()

    def take() -> bool:
        with budget_lock:
            if budget[0] is None:
                return True
            if budget[0] <= 0:
                return False
            budget[0] -= 1
            return True

    def feeder(slot: int, dev_index: int) -> None:
        tag = lambda *a: log(f"[gpu{dev_index}]", *a)
        try:
            if make_compute is None and native:
                share = None if max_tiles is None else (max_tiles + len(devices) - 1 - slot) // len(devices)
                if share is None or share > 0:
                    done[slot] = run_native(addr, port, dev_index, log=tag, senders=senders, max_tiles=share)
                return
            if make_compute is None:
                done[slot] = run_pipelined(addr, port, dev_index, log=tag, senders=senders, take=take)
                return
            compute = make_compute(dev_index)
            while not _stop_event.is_set() and take():
                if not do_workload_single(addr, port, compute=compute, log=tag):
                    break
                done[slot] += 1
        except BaseException as e:  # surfaced to the caller below
            errors.append(e)

    threads = [threading.Thread(target=feeder, args=(s, d), daemon=True) for s, d in enumerate(devices)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return done


_main_result: List[object] = []


def _run_farm_catching(addr, port, devices, native):
    try:
        return run_farm(addr, port, devices, native=native, senders=4)
    except BaseException as e:   # handed to the main thread
        return e


def main(argv: Optional[Sequence[str]] = None) -> None:
    """WorkerCUDA.py:178-184: prompts for the server address and port on stdin, then works until the
    server has nothing left.  Optional argv: ADDR PORT [gpu,gpu,...] to skip the prompts."""
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) >= 2:
        addr, port = argv[0], int(argv[1])
    else:
        addr = input("Server Addr> ")
        port = int(input("Server Port> "))
    devices = [int(x) for x in argv[2].split(",")] if len(argv) >= 3 else None
    # every listed GPU gets its own pipelined feeder (an explicit single index -- `worker ADDR PORT 3` --
    # runs on THAT GPU; round 1 sent it to GPU 0); no list = every visible GPU, and run_farm raises when there is none.
    # The feeders are the native loop (mbk_worker_run); `worker ADDR PORT GPUS python` keeps the Python one.
    # The native loop is one blocking C call per GPU: its sockets have timeouts (a dead server ends it with an error
    # instead of hanging it), and Ctrl-C asks it to stop leasing and drain (mbk_net_set_option(MBK_NET_STOP)).
    native = not (len(argv) >= 4 and argv[3] == "python")
    farm = threading.Thread(target=lambda: _main_result.append(_run_farm_catching(addr, port, devices, native)), daemon=True)
    _main_result.clear()
    farm.start()
    try:
        while farm.is_alive():
            farm.join(0.2)
    except KeyboardInterrupt:
        print("interrupt: finishing the tiles in flight, leasing no more")
        request_stop()      # native loops: MBK_NET_STOP; Python loops: the module's stop event
        farm.join()
    if _main_result and isinstance(_main_result[0], BaseException):
        raise _main_result[0]
    log_stats = dict(stats)
    print("tiles:", log_stats)


if __name__ == "__main__":
    main()
