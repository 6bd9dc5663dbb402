import torch

def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us

def row(name, us, flops, bytes_):
    print(f"{name:44s} {us:8.1f} us  {flops/us/1e6:7.1f} TFLOP/s  {bytes_/us/1e6:6.2f} TB/s", flush=True)

