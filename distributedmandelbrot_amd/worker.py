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

Differences from the reference worker, all wire-compatible:
  * `sendall` / receive-exactly instead of bare `send` / `recv(4)` (WorkerCUDA.py:104-107,168 may
    transfer short; the bytes on the wire are identical when nothing is cut short);
  * `run_farm`: one feeder thread per GPU, each an ordinary protocol client (the Distributer accepts
    any number of clients, Distributer.cs:226-297) -- the per-GPU work queue IS the Distributer's
    lease table, so there is no RCCL and no GPU<->GPU traffic;
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
    return (f"kernel {st.kernel_ms:.3f} ms, D2H {st.d2h_ms:.3f} ms, {st.pixel_iterations / 1e9:.2f} G pixel-iterations "
            f"({rate:.0f} G/s), {st.never_pixels} in-set pixels, stored as {kind}")


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        part = sock.recv(n - len(buf))
        if not part:
            raise ConnectionError(f"connection closed after {len(buf)} of {n} bytes")
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
    """First connection of WorkerCUDA.py:115-134.  None == 0x11 (no workload available)."""
    with socket.create_connection((addr, port), timeout=timeout) as sock:
        sock.sendall(struct.pack("B", REQUEST_CODE))
        response = _recv_exact(sock, 1)[0]
        if response == WORKLOAD_AVAILABLE_CODE:
            return receive_workload(sock)
        if response == WORKLOAD_NOT_AVAILABLE_CODE:
            return None
        raise Exception("Unknown response code to request: " + str(response))  # WorkerCUDA.py:131-132


def submit_workload(addr: str, port: int, workload: Workload, out: np.ndarray,
                    timeout: Optional[float] = None) -> bool:
    """Second connection of WorkerCUDA.py:148-172.  True == accepted and sent, False == 0x21."""
    payload = memoryview(np.ascontiguousarray(out, dtype=np.uint8)).cast("B")
    if len(payload) != CHUNK_BYTES:
        raise ValueError(f"tile payload must be {CHUNK_BYTES} bytes, got {len(payload)}")
    with socket.create_connection((addr, port), timeout=timeout) as sock:
        sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        # one segment for opcode + header: the server reads them with separate 100 ms-timeout
        # receives (Distributer.cs:17,243-245,400), so do not dribble them
        sock.sendall(struct.pack("<BIIII", RESPONSE_CODE, *workload))
        response = _recv_exact(sock, 1)[0]
        if response == WORKLOAD_REJECT_CODE:
            return False
        if response != WORKLOAD_ACCEPT_CODE:
            raise Exception("Unknown response code to request: " + str(response))  # WorkerCUDA.py:165-166
        try:
            sock.sendall(payload)  # exactly 16 777 216 raw bytes, no header (WorkerCUDA.py:168)
        except (ConnectionResetError, BrokenPipeError):
            # The reference server reads the payload with ONE Socket.Receive (Distributer.cs:416) and
            # then closes; with unread bytes in flight that close is a TCP reset.  By then it has
            # already marked the tile completed (:422-423).  The reference worker's single
            # `sock.send` never notices; neither must a drop-in.
            pass
    return True


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
    if submit_workload(addr, port, workload, out):
        log("Response accepted")
        log("Sent response")
    else:
        log("Response rejected")
    log("Process complete")
    return True


def run_farm(addr: str, port: int, devices: Optional[Sequence[int]] = None,
             make_compute: Optional[Callable[[int], ComputeFn]] = None,
             log: Callable[..., None] = print, max_tiles: Optional[int] = None) -> List[int]:
    """One feeder thread per GPU, each looping do_workload_single until the Distributer answers
    0x11.  Returns the number of tiles each feeder completed.  `make_compute(device_index)` builds
    the per-thread compute function (default: a MandelbrotDevice per GPU)."""
    if devices is None:
        from .device import device_count
        devices = list(range(device_count()))
        if not devices:
            raise RuntimeError("no gfx950 GPU visible and no CPU fallback exists")

    def default_make(dev_index: int) -> ComputeFn:
        from .device import MandelbrotDevice
        dev = MandelbrotDevice(dev_index)
        pinned = dev.pinned_empty((CHUNK_BYTES,), np.uint8)

        def compute(level, mrd, ir, ii):
            out, _, st = dev.datachunk(level, mrd, ir, ii, out_bytes=pinned)
            log(f"[gpu{dev_index}]", describe_stats(st))
            return out
        return compute

    make = make_compute or default_make
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
        try:
            compute = make(dev_index)
            while take():
                if not do_workload_single(addr, port, compute=compute,
                                          log=lambda *a: log(f"[gpu{dev_index}]", *a)):
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
    if devices is None:
        from .device import device_count
        n = device_count()
        devices = list(range(n))
    if len(devices) <= 1:
        while do_workload_single(addr, port):
            pass
    else:
        run_farm(addr, port, devices)


if __name__ == "__main__":
    main()
