#!/usr/bin/env python3
"""What the box allows between a file and HBM: pinned host->device copy rate (the ceiling of the GAF ingest), pread rate from the
page cache on k threads, and pageable->pinned memcpy rate.  usage: h2d_probe.py [GB]"""
import os, sys, time, tempfile, threading
import numpy as np, torch
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n = int(gb * (1 << 30))
pin = torch.empty(n, dtype=torch.uint8, pin_memory=True)
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
pin.fill_(7)
for chunk_mb in (16, 64, 256, int(gb * 1024)):
    c = chunk_mb << 20
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for off in range(0, n, c):
        dev[off:off + c].copy_(pin[off:off + c], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("pinned H2D, %4d MB chunks: %.1f GB/s" % (chunk_mb, n / dt / 1e9))
# two streams
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
c = 64 << 20
for k, off in enumerate(range(0, n, c)):
    with torch.cuda.stream(s1 if k & 1 else s2):
        dev[off:off + c].copy_(pin[off:off + c], non_blocking=True)
torch.cuda.synchronize(); print("pinned H2D, 64 MB chunks on two streams: %.1f GB/s" % (n / (time.perf_counter() - t0) / 1e9))
buf = pin.numpy()
for d in ("/dev/shm", tempfile.gettempdir()):
    try:
        fn = os.path.join(d, "h2d_probe.bin")
        with open(fn, "wb") as f: f.write(buf.tobytes())
        fd = os.open(fn, os.O_RDONLY)
        for k in (1, 4, 8, 16, 32, 64):
            piece = n // k
            def rd(i):
                off, end = i * piece, (i + 1) * piece
                mv = memoryview(buf)[off:end]
                done = 0
                while done < end - off:
                    r = os.preadv(fd, [mv[done:]], off + done)
                    if r <= 0: break
                    done += r
            t0 = time.perf_counter()
            th = [threading.Thread(target=rd, args=(i,)) for i in range(k)]
            [t.start() for t in th]; [t.join() for t in th]
            print("pread %s -> pinned, %2d threads: %.1f GB/s" % (d, k, n / (time.perf_counter() - t0) / 1e9))
        os.close(fd); os.unlink(fn)
    except Exception as e:
        print(d, "failed:", e)
src = np.frombuffer(bytearray(n), dtype=np.uint8)
for k in (1, 8, 16, 32):
    piece = n // k
    def cp(i): buf[i * piece:(i + 1) * piece] = src[i * piece:(i + 1) * piece]
    t0 = time.perf_counter()
    th = [threading.Thread(target=cp, args=(i,)) for i in range(k)]
    [t.start() for t in th]; [t.join() for t in th]
    print("memcpy pageable -> pinned, %2d threads: %.1f GB/s" % (k, n / (time.perf_counter() - t0) / 1e9))
