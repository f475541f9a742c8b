import torch, time
n = 92*1024*1024
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(f, reps=10):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps
def one(): 
    with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
def two():
    with torch.cuda.stream(s1): d[:n//2].copy_(h[:n//2], non_blocking=True)
    with torch.cuda.stream(s2): d[n//2:].copy_(h[n//2:], non_blocking=True)
def both_dir():
    with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
def d2h():
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
print("H2D one stream  %.1f GB/s" % (n/t(one)/1e9))
print("H2D two streams %.1f GB/s" % (n/t(two)/1e9))
print("D2H one stream  %.1f GB/s" % (n/t(d2h)/1e9))
print("H2D + D2H concurrently: %.1f GB/s each" % (n/t(both_dir)/1e9))
