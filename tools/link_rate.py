import torch, time
dev = torch.device("cuda", 0)
N = 1 << 30
h_up = torch.empty(N, dtype=torch.uint8).pin_memory(); h_dn = torch.empty(N, dtype=torch.uint8).pin_memory()
d_up = torch.empty(N, dtype=torch.uint8, device=dev); d_dn = torch.empty(N, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(up, dn, reps=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        if up:
            with torch.cuda.stream(s1): d_up.copy_(h_up, non_blocking=True)
        if dn:
            with torch.cuda.stream(s2): h_dn.copy_(d_dn, non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return reps * N * (int(up) + int(dn)) / dt / 1e9
run(True, True, 1)
print("H2D alone   %.1f GB/s" % run(True, False))
print("D2H alone   %.1f GB/s" % run(False, True))
print("both at once %.1f GB/s combined" % run(True, True))
