"""Host->device copy bandwidth of the box for several chunk sizes / stream counts (pinned memory)."""
import time

import torch


def run(chunk_mb, streams, total_mb=4096):
    n = chunk_mb * 1024 * 1024
    hosts = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(streams)]
    devs = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(streams)]
    ss = [torch.cuda.Stream() for _ in range(streams)]
    reps = max(1, total_mb // (chunk_mb * streams))
    for s, h, d in zip(ss, hosts, devs):
        with torch.cuda.stream(s):
            d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for s, h, d in zip(ss, hosts, devs):
            with torch.cuda.stream(s):
                d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return reps * streams * n / dt / 1e9


if __name__ == "__main__":
    for chunk, streams in [(10, 1), (10, 2), (10, 4), (10, 8), (64, 1), (64, 2), (256, 1), (1, 8), (2, 8)]:
        print("chunk %4d MB x %d streams : %6.1f GB/s" % (chunk, streams, run(chunk, streams)), flush=True)


def duplex(chunk_mb=10, up=8, down=2, seconds=1.0, busy_kernel=False):
    n = chunk_mb * 1024 * 1024
    hu = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(up)]
    du = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(up)]
    hd = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(down)]
    dd = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(down)]
    su = [torch.cuda.Stream() for _ in range(up)]
    sd = [torch.cuda.Stream() for _ in range(down)]
    sk = torch.cuda.Stream()
    big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda") if busy_kernel else None
    def one_round():
        for s, h, d in zip(su, hu, du):
            with torch.cuda.stream(s):
                d.copy_(h, non_blocking=True)
        for s, h, d in zip(sd, hd, dd):
            with torch.cuda.stream(s):
                h.copy_(d, non_blocking=True)
        if busy_kernel:
            with torch.cuda.stream(sk):
                big.add_(1.0)

    for _ in range(20):
        one_round()
    torch.cuda.synchronize()
    reps = 400
    t0 = time.perf_counter()
    for _ in range(reps):
        one_round()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return reps * up * n / dt / 1e9, reps * down * n / dt / 1e9
    for _ in range(reps):
        for s, h, d in zip(su, hu, du):
            with torch.cuda.stream(s):
                d.copy_(h, non_blocking=True)
        for s, h, d in zip(sd, hd, dd):
            with torch.cuda.stream(s):
                h.copy_(d, non_blocking=True)
        if busy_kernel:
            with torch.cuda.stream(sk):
                big.add_(1.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return reps * up * n / dt / 1e9, reps * down * n / dt / 1e9


if __name__ == "__main__":
    for up, down, busy in [(8, 0, False), (8, 2, False), (8, 8, False), (8, 2, True), (8, 0, True)]:
        u, d = duplex(10, up, max(down, 0) or 0, busy_kernel=busy) if down else duplex(10, up, 0, busy_kernel=busy)
        print("H2D x%d + D2H x%d%s : up %5.1f GB/s  down %5.1f GB/s" % (up, down, " + HBM-bound kernel" if busy else "", u, d), flush=True)
