"""Why a fed step loop can be slower than the same batches replayed (GPU box): the same 16 batches (a) cycled from HBM with no feed,
(b) cycled while a feed with live workers sits idle, (c) taken from the running feed, with host time per train_step call."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from coati_amd.data.feed import BatchFeed  # noqa: E402
from coati_amd.engine import Engine, ModelConfig  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    eng = Engine(ModelConfig(**bench.GRANDE), dev)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for name, (off, shape) in eng.layout.items():
            v = eng.view(name)
            if len(shape) == 2:
                v.copy_((torch.randn(shape, generator=g) * 0.02).to(dev))
            elif name.endswith("weight"):
                v.fill_(1.0)
    eng.refresh_shadows()
    vocab = json.load(open(os.path.join(ROOT, "tests", "golden", "tokenizer_real.json")))
    B = 1024
    up = torch.rand(B, device=dev) > 0.5
    tokens = int(os.environ.get("PROBE_TOKENS", "76"))
    kept = [{n: (v.clone() if v.is_cuda else v) for n, v in b.items()} for b in BatchFeed(bench._FeedPipe(vocab, B, 16, tokens=tokens), workers=4, device=dev)]
    print("shapes", [(int(b["raw_tokens"].shape[1]), int(b["tokens"].shape[1]), b["rows"].tolist()) for b in kept[:4]])

    def cycle(batches, n_rounds=2, label=""):
        for b in batches:
            eng.train_step(b, up, lr=5e-4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host = 0.0
        n = 0
        for _ in range(n_rounds):
            for b in batches:
                h0 = time.perf_counter()
                eng.train_step(b, up, lr=5e-4)
                host += time.perf_counter() - h0
                n += 1
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{label:44s} {dt * 1e3:8.2f} ms/step   host {host / n * 1e3:7.2f} ms per train_step call")

    rep = []
    for b in kept[:4]:
        for _ in range(2):
            eng.train_step(b, up, lr=5e-4)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            eng.train_step(b, up, lr=5e-4)
        torch.cuda.synchronize()
        rep.append((time.perf_counter() - t) / 3)
    print("replayed one at a time:", [round(x * 1e3, 2) for x in rep])
    cycle(kept, label="(a) 16 batches cycled, no feed")
    idle = BatchFeed(bench._FeedPipe(vocab, B, 400, tokens=tokens), workers=8, depth=3, device=dev)
    first = next(idle)          # workers alive, queues full, nobody consuming
    time.sleep(1.0)
    cycle(kept, label="(b) cycled, feed with 8 live workers idle")
    idle.close()
    time.sleep(0.5)
    cycle(kept, label="(a') cycled again, feed closed")
    def fed(label, workers, use_fed=True, nb=40):
        feed = BatchFeed(bench._FeedPipe(vocab, B, nb, tokens=tokens), workers=workers, depth=3, device=dev)
        k, host = 0, 0.0
        for b in feed:
            h0 = time.perf_counter()
            eng.train_step(b if use_fed else kept[k % 16], up, lr=5e-4)
            host += time.perf_counter() - h0
            k += 1
            if k == 8:
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                host = 0.0
        torch.cuda.synchronize()
        print(f"cap {eng._cap} growth events so far {eng.growth_events}")
        print(f"{label:44s} {(time.perf_counter() - t1) / (k - 8) * 1e3:8.2f} ms/step   host {host / (k - 8) * 1e3:7.2f} ms per train_step call; wait {feed.stats['wait_s']:.3f} get {feed.stats['get_s']:.3f} stage {feed.stats['stage_s']:.3f}")

    print(f"cap {eng._cap} growth events so far {eng.growth_events}")
    fed("(c0) fed, 8 workers, FIRST running feed", 8)
    fed("(c1) fed, inline batcher (no workers)", 0, nb=24)
    fed("(c2) feed running (8 workers), step on KEPT batches", 8, use_fed=False)
    fed("(c3) fed, 2 workers", 2)
    fresh = [{n: (v.clone() if v.is_cuda else v) for n, v in b.items()} for b in BatchFeed(bench._FeedPipe(vocab, B, 60, tokens=tokens), workers=8, device=dev)][16:]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in fresh:
        eng.train_step(b, up, lr=5e-4)
    torch.cuda.synchronize()
    print(f"{'(d) 44 NEW batches resident in HBM, no feed':44s} {(time.perf_counter() - t0) / len(fresh) * 1e3:8.2f} ms/step")
    t0 = time.perf_counter()
    for b in fresh:
        eng.train_step(b, up, lr=5e-4)
    torch.cuda.synchronize()
    print(f"{'(d2) the same 44 again':44s} {(time.perf_counter() - t0) / len(fresh) * 1e3:8.2f} ms/step")
    feed = BatchFeed(bench._FeedPipe(vocab, B, 40, tokens=tokens), workers=8, depth=3, device=dev)
    k, host = 0, 0.0
    for b in feed:
        h0 = time.perf_counter()
        eng.train_step(b, up, lr=5e-4)
        host += time.perf_counter() - h0
        k += 1
        if k == 8:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            host = 0.0
    torch.cuda.synchronize()
    print(f"{'(c) step loop on the running feed':44s} {(time.perf_counter() - t1) / (k - 8) * 1e3:8.2f} ms/step   host {host / (k - 8) * 1e3:7.2f} ms per train_step call; feed {feed.stats}")


if __name__ == "__main__":
    main()
