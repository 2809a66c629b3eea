# price of a fork/join inside a hipGraph: a chain of small kernels vs the same with some of them on a second branch
import torch, time
dev = torch.device("cuda")
x = torch.randn(64 * 1024, device=dev)
ys = [torch.empty_like(x) for _ in range(16)]
side = torch.cuda.Stream()

def chain(n):
    for i in range(n):
        torch.add(x, 1.0, out=ys[i])

def forked(n, k, pairs=1):
    # n kernels on the main branch, k on the side branch per pair
    per = n // pairs
    for p in range(pairs):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(k):
                torch.mul(x, 2.0, out=ys[8 + i])
        for i in range(per):
            torch.add(x, 1.0, out=ys[i])
        torch.cuda.current_stream().wait_stream(side)

def timeit(fn, reps=2000):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps): g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e6

for n in (4, 8):
    print(f"chain {n}: {timeit(lambda: chain(n)):.2f} us;  chain {n + 2}: {timeit(lambda: chain(n + 2)):.2f} us;  "
          f"{n} + 2 on a side branch: {timeit(lambda: forked(n, 2)):.2f} us;  {n} + 2x2 on two forks: {timeit(lambda: forked(n, 2, 2)):.2f} us")
