"""Where the host feed's time goes (run on the GPU box): per worker count, molecules/s and the feed thread's split into waiting on the
worker queues / staging + issuing copies, for device = cpu and cuda, with the main thread idle.  python tools/feed_probe.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from coati_amd.data.feed import BatchFeed  # noqa: E402

if __name__ == "__main__":
    vocab = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tokenizer_real.json")))
    B = 1024
    print("cores", os.cpu_count(), "torch threads", torch.get_num_threads(), "affinity", len(os.sched_getaffinity(0)))
    t = time.perf_counter()
    n = sum(1 for _ in bench._FeedPipe(vocab, B, 4)(0, 1))
    print(f"pipe alone in this process: {n * B / (time.perf_counter() - t):.0f} molecules/s")
    if torch.cuda.is_available():
        torch.zeros(1, device="cuda")          # the HIP context exists before any worker starts, as in the trainer
    for dev in ("cpu", "cuda") if torch.cuda.is_available() else ("cpu",):
        for w in (0, 1, 2, 4, 8, 16):
            nb = 6 if w <= 1 else 4 * max(w, 6)
            f = BatchFeed(bench._FeedPipe(vocab, B, nb), workers=w, depth=3, device=dev)
            t0 = time.perf_counter()
            k = 0
            for b in f:
                k += 1
                if k == 2:
                    t1 = time.perf_counter()
            dt = time.perf_counter() - t1
            print(f"{dev:5s} workers {w:2d}: {(k - 2) * B / dt:9.0f} molecules/s   feed thread: get {f.stats['get_s'] / k * 1e3:6.1f} ms/batch, stage {f.stats['stage_s'] / k * 1e3:6.1f} ms/batch")
