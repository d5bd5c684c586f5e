"""A Python stand-in for the reference's Distributer, used as the integration fixture because the
C# server cannot run here (no dotnet).  TEST INFRASTRUCTURE.

Restates the observable behaviour of Distributer.cs / DistributerWorkload.cs:
  * opcodes 0x00 request / 0x01 response; replies 0x10+workload / 0x11 and 0x20 / 0x21
    (Distributer.cs:30-45);
  * a request hands out the first tile that is neither completed nor under an unexpired lease,
    scanning levels x indexReal x indexImag in that order (Distributer.cs:335-353), sends the four
    u32 as four separate 4-byte sends (DistributerWorkload.cs:53-77), and registers a lease
    (Distributer.cs:374-376);
  * a response is accepted iff an unexpired lease matches all four fields
    (DistributerWorkload.cs:31-38,116-120; Distributer.cs:404); on accept the tile's 16 777 216 bytes
    are read and the lease moves to the completed set (:415-423);
  * one connection at a time, closed after each exchange (:226-297); unknown opcodes are logged and
    the connection closed (:266-268).
`faithful_single_receive=True` reproduces the reference defect of reading the payload with ONE
Receive call (Distributer.cs:416) -- whatever arrives in that call is kept, the rest stays zero.

"Reference-shaped" mode (round 4), for clients that open many connections at once: the loop is what the C# server's is
-- ONE accept thread that handles a connection to its end before accepting the next (Distributer.cs:226-297),
listen backlog 16 (:16,221) -- plus `receive_timeout=0.1` (the 100 ms ReceiveTimeout of :17,196-202),
`payload_seconds` (a floor on the time a 16 MiB payload takes: the real server needs >= 10 ms) and `rst_every=k`:
every k-th accepted connection is reset before a byte is read, which is what a client sees from a Windows/.NET host
whose backlog is full (Linux would drop the SYN instead).  `accepted` counts accepts, `resets_sent` the resets.
"""
from __future__ import annotations

import socket
import struct
import threading
import time
from typing import Dict, List, Optional, Tuple

import numpy as np

CHUNK_BYTES = 4096 * 4096
Workload = Tuple[int, int, int, int]


class FakeDistributer:
    def __init__(self, level_settings: List[Tuple[int, int]], lease_seconds: float = 3600.0,
                 faithful_single_receive: bool = False, receive_timeout: Optional[float] = 5.0,
                 payload_seconds: float = 0.0, rst_every: int = 0):
        self.level_settings = list(level_settings)
        self.lease_seconds = lease_seconds
        self.faithful_single_receive = faithful_single_receive
        self.receive_timeout = receive_timeout
        self.payload_seconds = payload_seconds
        self.rst_every = rst_every
        self.accepted = 0
        self.resets_sent = 0
        self.duplicate_completions = 0
        self.leases: List[Tuple[Workload, float]] = []
        self.completed: Dict[Workload, np.ndarray] = {}
        self.log: List[str] = []
        self.rejected: List[Workload] = []
        self._sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self._sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self._sock.bind(("127.0.0.1", 0))
        self._sock.listen(16)  # Distributer.cs:16
        self.port = self._sock.getsockname()[1]
        self._stop = False
        self._thread = threading.Thread(target=self._serve, daemon=True)
        self._thread.start()

    # -- helpers ---------------------------------------------------------------------------
    def expire_all_leases(self) -> None:
        self.leases = [(w, 0.0) for w, _ in self.leases]

    def wait_completed(self, n: int, timeout: float = 10.0) -> bool:
        """The worker returns as soon as its last byte is queued; wait for the server side."""
        end = time.monotonic() + timeout
        while len(self.completed) < n and time.monotonic() < end:
            time.sleep(0.005)
        return len(self.completed) >= n

    def close(self) -> None:
        self._stop = True
        try:
            socket.create_connection(("127.0.0.1", self.port), timeout=1).close()
        except OSError:
            pass
        self._thread.join(timeout=5)
        self._sock.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _live(self, w: Workload, now: float) -> bool:
        return any(lw == w and now < t for lw, t in self.leases)

    def _next_needed(self) -> Optional[Workload]:
        now = time.monotonic()
        for level, mrd in self.level_settings:
            for ir in range(level):
                for ii in range(level):
                    w = (level, mrd, ir, ii)
                    if w in self.completed or self._live(w, now):
                        continue
                    return w
        return None

    @staticmethod
    def _recv_exact(c: socket.socket, n: int) -> bytes:
        buf = bytearray()
        while len(buf) < n:
            part = c.recv(n - len(buf))
            if not part:
                raise ConnectionError("peer closed")
            buf += part
        return bytes(buf)

    # -- server loop -----------------------------------------------------------------------
    def _serve(self) -> None:
        while not self._stop:
            try:
                c, _ = self._sock.accept()
            except OSError:
                return
            if self._stop:
                c.close()
                return
            self.accepted += 1
            if self.rst_every and self.accepted % self.rst_every == 0:
                # what a full backlog looks like from a client of a Windows/.NET host: RST instead of SYN-ACK + service
                c.setsockopt(socket.SOL_SOCKET, socket.SO_LINGER, struct.pack("ii", 1, 0))
                c.close()
                self.resets_sent += 1
                continue
            c.settimeout(self.receive_timeout)
            try:
                op = self._recv_exact(c, 1)[0]
                if op == 0x00:
                    self._handle_request(c)
                elif op == 0x01:
                    self._handle_response(c)
                else:
                    self.log.append(f"unknown connection purpose {op}")
            except (socket.timeout, ConnectionError, OSError) as e:
                self.log.append(f"connection error: {e!r}")
            finally:
                c.close()

    def _handle_request(self, c: socket.socket) -> None:
        w = self._next_needed()
        if w is None:
            c.sendall(bytes([0x11]))
            return
        c.sendall(bytes([0x10]))
        for field in w:  # four separate 4-byte sends, DistributerWorkload.cs:62-75
            c.sendall(struct.pack("<I", field))
        self.leases.append((w, time.monotonic() + self.lease_seconds))

    def _handle_response(self, c: socket.socket) -> None:
        w = struct.unpack("<IIII", self._recv_exact(c, 16))
        now = time.monotonic()
        if not self._live(w, now):
            self.rejected.append(w)
            c.sendall(bytes([0x21]))
            return
        c.sendall(bytes([0x20]))
        if self.faithful_single_receive:
            data = bytearray(CHUNK_BYTES)
            got = c.recv_into(data, CHUNK_BYTES)  # ONE receive, Distributer.cs:416
            self.log.append(f"single receive got {got} bytes")
            payload = bytes(data)
        else:
            t0 = time.monotonic()
            payload = self._recv_exact(c, CHUNK_BYTES)
            if self.payload_seconds > 0:
                time.sleep(max(0.0, self.payload_seconds - (time.monotonic() - t0)))
        if w in self.completed:
            self.duplicate_completions += 1
        for k, (lw, t) in enumerate(self.leases):
            if lw == w and now < t:
                del self.leases[k]
                break
        self.completed[w] = np.frombuffer(payload, dtype=np.uint8).copy()
