"""Which part of staging a batch slows down when worker processes exist, per start method (fork / spawn / forkserver); GPU box."""
import os, sys, time, json, torch, numpy as np, multiprocessing as mp
sys.path.insert(0, os.getcwd())
def child(q, ev):
    a = np.zeros((1024, 114), dtype=np.int64)
    t_end = time.time() + 20.0          # never outlive the probe
    while not ev.is_set() and time.time() < t_end:
        t=time.time()
        while time.time()-t < 0.02: pass     # busy 20 ms
        try: q.put({"a": a, "b": a, "c": a}, timeout=0.1)
        except Exception: pass
    os._exit(0)
def run(nchild, method="fork"):
    ctx = mp.get_context(method)
    ev = ctx.Event(); qs=[]; ps=[]
    for i in range(nchild):
        q = ctx.Queue(maxsize=3); p = ctx.Process(target=child, args=(q, ev), daemon=True); p.start(); qs.append(q); ps.append(p)
    st = torch.cuda.Stream()
    pins = [{k: torch.empty(1024*114, dtype=torch.int64).pin_memory() for k in "abc"} for _ in range(4)]
    evs = [None]*4
    T = dict(get=0, sync=0, cp=0, to=0, rec=0)
    local = {k: np.zeros((1024,114), dtype=np.int64) for k in "abc"}
    N=40
    for i in range(N):
        t0=time.perf_counter()
        b = qs[i % nchild].get() if nchild else local
        t1=time.perf_counter()
        s = i % 4
        if evs[s] is not None: evs[s].synchronize()
        t2=time.perf_counter()
        for k in "abc": pins[s][k].numpy()[:] = b[k].reshape(-1)
        t3=time.perf_counter()
        with torch.cuda.stream(st):
            out = [pins[s][k].to("cuda", non_blocking=True) for k in "abc"]
            t4=time.perf_counter()
            e = torch.cuda.Event(); e.record(st)
        evs[s]=e
        t5=time.perf_counter()
        T["get"]+=t1-t0; T["sync"]+=t2-t1; T["cp"]+=t3-t2; T["to"]+=t4-t3; T["rec"]+=t5-t4
    torch.cuda.synchronize()
    print(method, nchild, {k: round(v/N*1e3,3) for k,v in T.items()})
    ev.set()
    for q in qs:
        try:
            while True: q.get_nowait()
        except Exception: pass
    for p in ps: p.join(timeout=1)
if __name__ == "__main__":
    torch.zeros(1, device="cuda")
    for m in ("fork", "spawn", "forkserver"):
        for n in (0, 4, 8):
            run(n, m)
    os._exit(0)
