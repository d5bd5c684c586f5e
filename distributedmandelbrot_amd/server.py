"""Stand-ins for the reference's two server loops, faithful to their wire and lease semantics, for
deployments (and tests) where the C# server is not available.  Host-side Python; nothing here touches
the GPU.  

  Distributer  <- Distributer.cs / DistributerWorkload.cs: opcodes 0x00/0x01, replies 0x10+workload /
                  0x11 and 0x20 / 0x21 (:30-45); first tile that is neither completed nor under an
                  unexpired lease, scanning levels x indexReal x indexImag (:335-353); 1 h leases
                  (:22,374-376); a response is accepted iff an unexpired lease matches (:404);
                  completed set reloaded from the store at start (:124,165-175).
                  Differences: payloads are read with receive-exactly (the reference's single
                  Socket.Receive, :416, may truncate a tile), and each connection is handled on its own
                  thread so that 8 GPU feeders do not serialise on Accept (:226-297).
  DataServer   <- DataServer.cs:156-224: request u32 level, indexReal, indexImag; reply u8 status
                  (0 accepted, 1 rejected: index >= level, 2 not available), then u32 length and the
                  chunk as DataChunk.Serialize writes it (:204-220).
"""
from __future__ import annotations

import socket
import struct
import threading
import time
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .chunkstore import CHUNK_BYTES, ChunkStore

Workload = Tuple[int, int, int, int]
DEFAULT_DISTRIBUTER_PORT = 59010  # Program.cs:13
DEFAULT_DATA_SERVER_PORT = 59011  # Program.cs:14


def _recv_exact(c: socket.socket, n: int, buf: Optional[bytearray] = None) -> bytearray:
    if buf is None:
        buf = bytearray(n)
    view, got = memoryview(buf), 0
    while got < n:
        k = c.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError(f"peer closed after {got} of {n} bytes")
        got += k
    return buf


class _TcpLoop:
    def __init__(self, host: str, port: int, backlog: int = 16):
        self._sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self._sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self._sock.bind((host, port))
        self._sock.listen(backlog)
        self.port = self._sock.getsockname()[1]
        self._stop = False
        self.errors: List[str] = []
        self._thread = threading.Thread(target=self._accept_loop, daemon=True)
        self._thread.start()

    def _accept_loop(self) -> None:
        while not self._stop:
            try:
                c, _ = self._sock.accept()
            except OSError:
                return
            if self._stop:
                c.close()
                return
            threading.Thread(target=self._client, args=(c,), daemon=True).start()

    def _client(self, c: socket.socket) -> None:
        try:
            c.settimeout(30.0)
            self.handle(c)
        except (socket.timeout, ConnectionError, OSError) as e:  # Distributer.cs:279-290: log, carry on
            self.errors.append(repr(e))
        finally:
            c.close()

    def handle(self, c: socket.socket) -> None:  # pragma: no cover - abstract
        raise NotImplementedError

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


class Distributer(_TcpLoop):
    def __init__(self, level_settings: Sequence[Tuple[int, int]], store: Optional[ChunkStore] = None,
                 host: str = "127.0.0.1", port: int = 0, lease_seconds: float = 3600.0):
        self.level_settings = [(int(l), int(m)) for l, m in level_settings]
        self.store = store
        self.lease_seconds = lease_seconds
        self._state = threading.Lock()
        self.leases: List[Tuple[Workload, float]] = []
        self.receiving: dict = {}   # leases claimed by a response whose payload is still arriving
        # completed is keyed without mrd, like the index (Distributer.cs:165-175)
        self.completed = set(store.completed()) if store is not None else set()
        self.rejected: List[Workload] = []
        self.received = 0
        # Payload buffers are recycled: a fresh bytearray(16 MiB) per tile is a 16 MiB zero-fill plus 4 096 page
        # faults, ~2 ms -- more than the GPU needs for most tiles.  A store that keeps a payload must copy it
        # (ChunkStore.save_chunk writes it out before returning).
        self._pool: List[bytearray] = []
        super().__init__(host, port)

    def _take_buffer(self) -> bytearray:
        with self._state:
            if self._pool:
                return self._pool.pop()
        return bytearray(CHUNK_BYTES)

    def _give_buffer(self, buf: bytearray) -> None:
        with self._state:
            if len(self._pool) < 16:
                self._pool.append(buf)

    def _next_needed(self) -> Optional[Workload]:
        now = time.monotonic()
        self.leases = [(w, t) for w, t in self.leases if now < t]  # the 5-minute sweeper, :153-160
        leased = {w for w, _ in self.leases} | set(self.receiving)
        for level, mrd in self.level_settings:
            for ir in range(level):
                for ii in range(level):
                    w = (level, mrd, ir, ii)
                    if (level, ir, ii) in self.completed or w in leased:
                        continue
                    return w
        return None

    def handle(self, c: socket.socket) -> None:
        op = _recv_exact(c, 1)[0]
        if op == 0x00:
            with self._state:
                w = self._next_needed()
                if w is not None:
                    self.leases.append((w, time.monotonic() + self.lease_seconds))
            if w is None:
                c.sendall(bytes([0x11]))
            else:
                c.sendall(bytes([0x10]) + struct.pack("<IIII", *w))
        elif op == 0x01:
            w = struct.unpack("<IIII", _recv_exact(c, 16))
            now = time.monotonic()
            # Claim the lease in the same critical section that tests it: the reference handles one
            # connection at a time (Distributer.cs:226-297), so of two responses for one tile (an expired
            # worker and its successor) the second is rejected (:404-423).  With a thread per connection
            # the claim must be explicit, or both would be accepted and stored.
            with self._state:
                claimed = None
                for k, (lw, t) in enumerate(self.leases):
                    if lw == w and now < t:
                        claimed = self.leases.pop(k)
                        self.receiving[w] = claimed
                        break
            if claimed is None:
                self.rejected.append(w)
                c.sendall(bytes([0x21]))
                return
            raw = self._take_buffer()
            try:
                c.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 4 << 20)   # fewer, larger receives
                c.sendall(bytes([0x20]))
                payload = np.frombuffer(_recv_exact(c, CHUNK_BYTES, raw), dtype=np.uint8)
            except BaseException:
                with self._state:       # the tile did not arrive: the lease is live again
                    self.receiving.pop(w, None)
                    self.leases.append(claimed)
                raise
            with self._state:
                self.receiving.pop(w, None)
                self.completed.add((w[0], w[2], w[3]))
            if self.store is not None:  # the reference saves on a thread-pool task (Distributer.cs:436-442)
                self.store.save_chunk(w[0], w[2], w[3], payload)
            del payload
            self._give_buffer(raw)
            with self._state:
                self.received += 1      # counts tiles that are accepted AND stored
        else:
            self.errors.append(f"unknown connection purpose {op}")

    def all_done(self) -> bool:
        with self._state:
            return self._next_needed() is None and not self.leases and not self.receiving


class DataServer(_TcpLoop):
    ACCEPTED, REJECTED, NOT_AVAILABLE = 0x00, 0x01, 0x02  # DataServer.cs:15-20

    def __init__(self, store: ChunkStore, host: str = "127.0.0.1", port: int = 0):
        self.store = store
        super().__init__(host, port)

    def handle(self, c: socket.socket) -> None:
        level, ir, ii = struct.unpack("<III", _recv_exact(c, 12))
        if ir >= level or ii >= level:
            c.sendall(bytes([self.REJECTED]))
            return
        entry = self.store.find(level, ir, ii)
        if entry is None:
            c.sendall(bytes([self.NOT_AVAILABLE]))
            return
        stream = self.store.load_serialized(entry)
        c.sendall(bytes([self.ACCEPTED]) + struct.pack("<I", len(stream)))
        c.sendall(stream)
