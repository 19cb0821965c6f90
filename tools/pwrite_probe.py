#!/usr/bin/env python3
"""Probe for config C5's output files: does ONE 132 MB file (an absorption field of the io grid) go into the page cache
faster from several threads (pwrite to disjoint ranges)? On the GPU box's ext4: no -- 16 ms at 1, 2, 4 and 8 threads
(buffered writes to one file are serialised on its inode), fresh file or overwritten. Hence the sweep's ring of host
sets: what scales is writing DIFFERENT files side by side (drivers --pipeline=K). Usage: python tools/pwrite_probe.py"""
import os, sys, time, threading, numpy as np, tempfile
n = 132 << 20
buf = np.random.randint(0, 255, n, dtype=np.uint8)
mv = memoryview(buf)
d = tempfile.mkdtemp(dir=os.environ.get("PW_DIR", "/tmp"))
def run(nt, fresh):
    path = os.path.join(d, f"f_{nt}_{fresh}_{time.time_ns()}.bin" if fresh else f"f_{nt}.bin")
    t0 = time.perf_counter()
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    os.ftruncate(fd, n)
    chunk = n // nt
    def w(i):
        off = i * chunk; end = n if i == nt - 1 else off + chunk
        while off < end:
            off += os.pwrite(fd, mv[off:min(end, off + (8 << 20))], off)
    ts = [threading.Thread(target=w, args=(i,)) for i in range(nt)]
    [t.start() for t in ts]; [t.join() for t in ts]
    os.close(fd)
    return time.perf_counter() - t0
import subprocess
print(subprocess.run(["df", "-T", d], capture_output=True, text=True).stdout)
for fresh in (True, False):
    for nt in (1, 2, 4, 8):
        ts = [run(nt, fresh) for _ in range(4)]
        print("fresh" if fresh else "overwrite", nt, "threads: %.1f ms  %.1f GB/s" % (min(ts) * 1e3, n / min(ts) / 1e9), flush=True)
