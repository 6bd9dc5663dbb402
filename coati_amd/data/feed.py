"""The host feed in front of the training step: rows -> batches -> clip_ar_xform -> device, ahead of the step that uses them.

What it replaces: coati/data/batch_pipe.py:78-131 (`UrBatcher`: filter rows by required fields / md5 rank shard / partition,
collect `batch_size` of them, `stack_batch`, `xform_routine`) iterated SERIALLY inside the step loop (train_coati.py:363-376),
the xform moving every tensor to the GPU with a pageable, synchronous `.to(device)` (clip_e2e.py:288-300).  One host thread
running `clip_ar_xform` is slower than the MI355X engine (33-45 k molecules/s at ~24 tokens per row against 50 k), so here

* `UrBatcher` keeps the reference's row policy and batch contents but can materialise only every n-th batch (`worker`,
  `n_workers`): every worker walks the same filtered row stream (an md5 per row), stacks + transforms only its own batches;
* `BatchFeed` runs the workers as processes behind bounded queues, takes the batches back in batch order (so the stream is the
  SAME for any worker count, and for workers = 0 = inline), copies them into a ring of pinned staging buffers and issues the
  H2D copies `non_blocking` on a copy stream; the consumer's stream waits on the copy's event only.
* per-batch seeding (`seed`): Python / numpy / torch generators of the producing process are re-seeded from
  (seed, rank, partition, batch index) before `xform_routine` runs, so the augmentation draws do not depend on which worker made the batch.

Workers never touch the GPU (xform_routine must build CPU tensors: pass device="cpu" to clip_ar_xform), and they are NOT forked
from the training process: children forked from a process that holds a HIP context slow that process's own HIP calls down for as
long as they live (hipEventSynchronize on a finished copy 0.002 -> 4-9 ms, the training step 2.5 x slower:
`tools/probes/feed_stage_probe.py`, `profiles/r06_feed_probe.txt`).  They come from a fork server that is itself a fresh interpreter
(`multiprocessing` "forkserver", this module preloaded so that a worker starts in milliseconds); `make_batcher` therefore has to
pickle (a module-level class instance; `TrieTokenizer` pickles as its vocabulary)."""
import hashlib
import os
import queue
import random
import threading
import time
from typing import Any, Callable, Dict, Iterable, Iterator, List, Optional

import numpy as np
import torch

from .batch_pipe import get_mod_from_str, stack_batch


def batch_seed(seed: int, rank: int, partition: str, index: int) -> int:
    """64-bit seed of one batch's augmentation draws: a hash, so neighbouring batches / ranks share no generator state"""
    h = hashlib.blake2b(f"{seed}/{rank}/{partition}/{index}".encode(), digest_size=8).digest()
    return int.from_bytes(h, "little")


def seed_everything(s: int):
    """the HOST generators clip_ar_xform draws from (`random`; numpy / torch's CPU generator for coord_noise).  Not the device
    generators: with workers = 0 this runs in the feed thread of the training process, whose use_point draws come from the GPU's"""
    random.seed(s)
    np.random.seed(s & 0xFFFFFFFF)
    torch.default_generator.manual_seed(s & 0x7FFFFFFFFFFFFFFF)


class UrBatcher:
    """batch_pipe.py:78-131 with the reference's arguments (same row filters in the same order, same batch boundaries,
    `skip_last`), plus `worker` / `n_workers` (materialise batches with index % n_workers == worker only; the others cost
    the row filters and nothing else) and `seed` (per-batch re-seeding, see the module docstring).  Yields
    (batch_index, xform_routine(stack_batch(rows)))."""

    def __init__(self, dp: Iterable[Dict[str, Any]], batch_size: int = 32, partition: str = "raw", xform_routine=lambda X: X,
                 partition_routine=lambda X: ["raw", "train", "test"], distributed_rankmod_total=None, distributed_rankmod_rank=1,
                 direct_mode=False, required_fields=(), skip_last=True, worker: int = 0, n_workers: int = 1, seed: Optional[int] = None):
        self.dp, self.batch_size, self.partition = dp, int(batch_size), partition
        self.xform_routine, self.partition_routine = xform_routine, partition_routine
        self.distributed_rankmod_total, self.distributed_rankmod_rank = distributed_rankmod_total, distributed_rankmod_rank
        self.direct_mode, self.required_fields, self.skip_last = direct_mode, list(required_fields), skip_last
        self.worker, self.n_workers, self.seed = int(worker), max(1, int(n_workers)), seed

    def _emit(self, index, rows):
        if self.seed is not None:
            rank = self.distributed_rankmod_rank if self.distributed_rankmod_total is not None else 0
            seed_everything(batch_seed(self.seed, rank, self.partition, index))
        return index, self.xform_routine(stack_batch(rows, return_coords=True))

    def __iter__(self):
        rows: List[Dict[str, Any]] = []
        index, mine, kept = 0, self.worker == 0, 0
        for row in self.dp:
            if not all(k in row for k in self.required_fields):
                continue
            mod = row["mod_molecule"] = get_mod_from_str(row["smiles"], 100_000)
            if self.distributed_rankmod_total is not None and (mod % self.distributed_rankmod_total) != self.distributed_rankmod_rank:
                continue
            if self.partition not in self.partition_routine(row):
                continue
            if mine:
                rows.append(row)
            kept += 1
            if kept == self.batch_size:
                if mine:
                    yield self._emit(index, rows)
                index, kept, rows = index + 1, 0, []
                mine = (index % self.n_workers) == self.worker
        if kept and not self.skip_last and mine:
            yield self._emit(index, rows)


def _tensors_only(batch: Dict[str, Any]) -> Dict[str, torch.Tensor]:
    """what crosses the process boundary and the PCIe link: the tensors of the batch (object columns such as the SMILES strings
    stay in the worker; `rows` = the packed-row counts, a host tensor the engine reads on the host)"""
    return {k: v for k, v in batch.items() if isinstance(v, torch.Tensor)}


def _worker_main(make_batcher, worker, n_workers, out_q, stop_ev):
    try:
        torch.set_num_threads(1)
        for index, batch in make_batcher(worker, n_workers):
            # numpy over the pipe (one pickle of ~ 2 MB per batch): torch's fd-passing reduction needs the producer alive when
            # the consumer unpickles, which a worker that has finished its stream is not
            item = (index, {k: v.numpy() for k, v in _tensors_only(batch).items()})
            while not stop_ev.is_set():
                try:
                    out_q.put(item, timeout=0.1)
                    break
                except queue.Full:
                    continue
            if stop_ev.is_set():
                break
        out_q.put((-1, None))
    except BaseException as ex:   # the consumer re-raises
        import traceback
        out_q.put((-2, f"{type(ex).__name__}: {ex}\n{traceback.format_exc()}"))
    finally:
        out_q.close()
        out_q.join_thread()


class _Slot:
    """one set of pinned staging buffers + one set of device buffers (both grown on demand, kept: no allocator traffic per batch),
    the event of the last H2D copy into them, and the consumer's "done with it" event"""

    def __init__(self):
        self.buf: Dict[str, torch.Tensor] = {}
        self.dev: Dict[str, torch.Tensor] = {}
        self.event = None
        self.done = None          # (batch number, event recorded by the consumer when it asked for the next batch)

    def device_view(self, name, t: torch.Tensor, device) -> torch.Tensor:
        n = t.numel()
        d = self.dev.get(name)
        if d is None or d.dtype != t.dtype or d.numel() < n:
            d = torch.empty(max(n + n // 4, 1), dtype=t.dtype, device=device)
            self.dev[name] = d
        return d[:n].view(t.shape)

    def stage(self, name, t: torch.Tensor) -> torch.Tensor:
        n = t.numel()
        b = self.buf.get(name)
        if b is None or b.dtype != t.dtype or b.numel() < n:
            b = torch.empty(max(n, 1), dtype=t.dtype).pin_memory()
            self.buf[name] = b
        # one memcpy through numpy: a torch copy_ of a few MB fans out over the intra-op pool, whose threads then spin next to the
        # worker processes (measured on the 256-core box: 16-30 ms per batch in copy_, 0.3 ms as a memcpy: tools/feed_probe.py)
        b.numpy()[:n] = t.contiguous().view(-1).numpy()
        return b[:n].view(t.shape)


class BatchFeed:
    """Iterator over device batches, produced ahead of their use.

    make_batcher(worker, n_workers) -> iterable of (batch_index, batch dict of CPU tensors) holding the batches with
    index % n_workers == worker in increasing order (a `UrBatcher`); picklable (see the module docstring) unless workers = 0.
    workers = 0: the batcher runs inline in the feed thread (still ahead of the step, still pinned + asynchronous H2D).
    depth: batches in flight per worker queue and on the device side.  host_keys: entries left on the host (`rows`).
    device = "cpu": no staging, no streams (CPU tests of ordering and determinism).

    The device tensors of a batch are views into a ring of depth + 4 preallocated slots (no allocation per batch: a per-batch
    allocation from the feed thread costs the training stream ~ 25 ms per step, tools/feed_e2e_probe.py): a batch is valid until
    the NEXT call of __next__ -- work already enqueued on the consumer's stream at that moment is safe, later use needs a clone."""

    def __init__(self, make_batcher: Callable[[int, int], Iterable], workers: int = 2, depth: int = 3, device="cuda",
                 host_keys=("rows",), mp_context: Optional[str] = None):
        self.make_batcher, self.workers, self.depth = make_batcher, max(0, int(workers)), max(2, int(depth))
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.host_keys = set(host_keys)
        self.ctx_name = mp_context or os.environ.get("COATI_FEED_MP", "forkserver")
        # wait_s: the consumer blocked in __next__; get_s: the feed thread blocked on the worker queues (or, inline, making the batch);
        # stage_s: pinned staging + issuing the copies
        self.stats = {"batches": 0, "wait_s": 0.0, "h2d_bytes": 0, "get_s": 0.0, "stage_s": 0.0}
        self._procs, self._queues, self._thread = [], [], None
        self._ready: "queue.Queue" = queue.Queue(maxsize=self.depth)
        self._stop = threading.Event()
        self._started = False
        self._last, self._taken = None, 0

    # ---- producer side -------------------------------------------------------------------------------------------------------
    def _start(self):
        self._started = True
        if self.workers > 0:
            import multiprocessing as mp
            ctx = mp.get_context(self.ctx_name)
            if self.ctx_name == "forkserver":
                ctx.set_forkserver_preload(["coati_amd.data.feed", "coati_amd.models.encoding.clip_e2e"])
            self._mp_stop = ctx.Event()
            for w in range(self.workers):
                q = ctx.Queue(maxsize=self.depth)
                p = ctx.Process(target=_worker_main, args=(self.make_batcher, w, self.workers, q, self._mp_stop), daemon=True)
                p.start()
                self._procs.append(p)
                self._queues.append(q)
        if self.device.type == "cuda":
            self._copy_stream = torch.cuda.Stream(device=self.device)
            self._slots = [_Slot() for _ in range(self.depth + 4)]
        self._thread = threading.Thread(target=self._pump, name="coati-feed", daemon=True)
        self._thread.start()

    def _host_batches(self) -> Iterator[Dict[str, torch.Tensor]]:
        """the batches in batch-index order: worker queues are read round-robin (worker w holds indices w, w + n, ...)"""
        if self.workers == 0:
            for _, b in self.make_batcher(0, 1):
                yield _tensors_only(b)
            return
        live = [True] * self.workers
        w = 0
        while any(live):
            if live[w]:
                while True:
                    try:
                        index, b = self._queues[w].get(timeout=0.2)
                        break
                    except queue.Empty:
                        if self._stop.is_set():
                            return
                        if not self._procs[w].is_alive() and self._queues[w].empty():
                            raise RuntimeError(f"feed worker {w} died without a result")
                if index == -2:
                    raise RuntimeError(f"feed worker {w} failed: {b}")
                if index == -1:
                    live[w] = False
                    # a worker ends only when the row stream does: the workers behind it in the rotation have at most their own
                    # final batches left, which come out in order as the rotation continues
                else:
                    yield {k: torch.from_numpy(v) for k, v in b.items()}
            w = (w + 1) % self.workers

    def _to_device(self, k: int, b: Dict[str, torch.Tensor]):
        if self.device.type != "cuda":
            return b, None
        R = len(self._slots)
        slot = self._slots[k % R]
        if slot.event is not None:
            slot.event.synchronize()          # the copy that last read this slot's pinned buffers has finished
        out = {}
        with torch.cuda.stream(self._copy_stream):
            if k >= R:
                # the slot's device buffers held batch k - R: the copy stream waits (on the device) for the consumer's kernels that
                # read it.  The consumer marks a batch done when it asks for the next one, and it is >= 2 batches past k - R by now
                # (queue depth + the one in hand < R - 2); the loop below only guards that invariant.
                while not (slot.done is not None and slot.done[0] == k - R):
                    if self._stop.is_set():
                        return None, None
                    time.sleep(0.0005)
                self._copy_stream.wait_event(slot.done[1])
            for name, t in b.items():
                if name in self.host_keys:
                    out[name] = t
                    continue
                d = slot.device_view(name, t, self.device)
                d.copy_(slot.stage(name, t), non_blocking=True)
                out[name] = d
                self.stats["h2d_bytes"] += t.numel() * t.element_size()
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        slot.event = ev
        return out, ev

    def _pump(self):
        try:
            torch.set_num_threads(1)          # this thread's OpenMP setting only: the inline batcher's small tensor ops must not fan out
            if self.device.type == "cuda":
                torch.cuda.set_device(self.device)
            it = self._host_batches()
            k = 0
            while True:
                t0 = time.perf_counter()
                b = next(it, None)
                t1 = time.perf_counter()
                if b is None:
                    break
                item = self._to_device(k, b)
                k += 1
                self.stats["get_s"] += t1 - t0
                self.stats["stage_s"] += time.perf_counter() - t1
                while not self._stop.is_set():
                    try:
                        self._ready.put(item, timeout=0.1)
                        break
                    except queue.Full:
                        continue
                if self._stop.is_set():
                    return
            self._ready.put((None, None))
        except BaseException as ex:
            self._ready.put((ex, "error"))

    # ---- consumer side -------------------------------------------------------------------------------------------------------
    def __iter__(self):
        return self

    def __next__(self) -> Dict[str, torch.Tensor]:
        if not self._started:
            self._start()
        if self._last is not None:
            # everything enqueued so far that reads the previous batch is in front of this event: its slot may be refilled behind it
            k, slot = self._last
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            slot.done = (k, done)
            self._last = None
        t0 = time.perf_counter()
        b, ev = self._ready.get()
        self.stats["wait_s"] += time.perf_counter() - t0
        if ev == "error":
            self.close()
            raise b
        if b is None:
            self.close()
            raise StopIteration
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            self._last = (self._taken, self._slots[self._taken % len(self._slots)])
        self._taken += 1
        self.stats["batches"] += 1
        return b

    def close(self):
        self._stop.set()
        if self._procs:
            self._mp_stop.set()
            for q in self._queues:            # unblock producers stuck in put()
                try:
                    while True:
                        q.get_nowait()
                except Exception:
                    pass
            for p in self._procs:
                p.join(timeout=2.0)
                if p.is_alive():
                    p.terminate()
            self._procs, self._queues = [], []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- a row source with the reference's row format, for the bench / tests (the 340 GB corpus is not here) ------------------------
class SyntheticRows:
    """Rows {"smiles", "source_collection", "atoms" [n], "coords" [n, 3]} like the dataset's unstacked pickles
    (coati/data/dataset.py, batch_pipe.py:108-131), generated per row index from `seed`: the "SMILES" is a concatenation of
    vocabulary entries that tokenises to ~ `tokens` ids (uniform in [tokens/2, tokens]).  Iterating is cheap next to the xform
    (that is the point: the feed measures stack_batch + clip_ar_xform + H2D, not the generator)."""

    def __init__(self, smiles_tokens: List[str], n_rows: int, tokens: int = 76, atoms: int = 16, seed: int = 0,
                 collections=("geom_drugs", "chembl_mols")):
        # single-character-safe pieces: a concatenation of multi-character vocabulary entries can re-segment under longest match,
        # which only shifts the realised length a little
        self.vocab = [t for t in smiles_tokens if t and not t.startswith("[")]
        if not self.vocab:
            self.vocab = list(smiles_tokens)
        self.n_rows, self.tokens, self.atoms, self.seed, self.collections = int(n_rows), int(tokens), int(atoms), int(seed), list(collections)
        self.species = np.array([1, 6, 7, 8, 9, 16, 17])

    BLOCK = 256        # rows drawn per generator (one generator per row would cost as much as tokenising the row)

    def block(self, k: int) -> List[Dict[str, Any]]:
        n = min(self.BLOCK, self.n_rows - k * self.BLOCK)
        g = np.random.default_rng((self.seed, k))
        n_tok = g.integers(max(1, self.tokens // 2), self.tokens + 1, size=n)
        ids = g.integers(0, len(self.vocab), size=(n, self.tokens))
        n_at = g.integers(max(1, self.atoms // 2), self.atoms + 1, size=n)
        z = self.species[g.integers(0, len(self.species), size=(n, self.atoms))].astype(np.float64)
        xyz = g.normal(0.0, 1.5, size=(n, self.atoms, 3))
        col = g.integers(0, len(self.collections), size=n)
        v = self.vocab
        return [{"smiles": "".join([v[j] for j in ids[r, : n_tok[r]]]), "source_collection": self.collections[col[r]],
                 "atoms": z[r, : n_at[r]], "coords": xyz[r, : n_at[r]]} for r in range(n)]

    def __len__(self):
        return self.n_rows

    def __iter__(self):
        for k in range((self.n_rows + self.BLOCK - 1) // self.BLOCK):
            yield from self.block(k)
