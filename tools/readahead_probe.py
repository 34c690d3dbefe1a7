#!/usr/bin/env python3
"""Does asynchronous read-ahead work on this file system?  One 1 GiB file, evicted before each trial; time to read it
after (a) nothing, (b) posix_fadvise(WILLNEED) + a pause, (c) readahead(2) + a pause, (d) a helper thread reading it."""
import ctypes
import os
import sys
import tempfile
import threading
import time

d = tempfile.mkdtemp(prefix="ra_probe_", dir=os.environ.get("KWAGE_PROBE_DIR", "/tmp"))
p = os.path.join(d, "f.bin")
size = 1 << 30
with open(p, "wb") as fh:
    fh.write(os.urandom(1 << 20) * (size >> 20))
libc = ctypes.CDLL("libc.so.6", use_errno=True)
libc.readahead.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_size_t]
buf = bytearray(8 << 20)


def evict():
    fd = os.open(p, os.O_RDONLY)
    os.fsync(fd)
    os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
    os.close(fd)


def read_all():
    t0 = time.perf_counter()
    with open(p, "rb", buffering=0) as fh:
        while fh.readinto(buf):
            pass
    return time.perf_counter() - t0


evict(); print("evicted, read:                         %.3f s" % read_all())
print("cached, read:                          %.3f s" % read_all())
evict()
fd = os.open(p, os.O_RDONLY)
t0 = time.perf_counter(); os.posix_fadvise(fd, 0, size, os.POSIX_FADV_WILLNEED); t_call = time.perf_counter() - t0
time.sleep(0.5)
print("fadvise(WILLNEED) took %.3f s; after 0.5 s, read: %.3f s" % (t_call, read_all()))
os.close(fd)
evict()
fd = os.open(p, os.O_RDONLY)
t0 = time.perf_counter()
for off in range(0, size, 32 << 20):
    libc.readahead(fd, off, 32 << 20)
t_call = time.perf_counter() - t0
time.sleep(0.5)
print("readahead() x32 took %.3f s; after 0.5 s, read: %.3f s" % (t_call, read_all()))
os.close(fd)
evict()
t = threading.Thread(target=read_all); t0 = time.perf_counter(); t.start(); t.join()
print("helper thread reading:                 %.3f s; then read: %.3f s" % (time.perf_counter() - t0, read_all()))
os.remove(p); os.rmdir(d)
