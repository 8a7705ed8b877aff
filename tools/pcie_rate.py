# raw host <-> device copy rates of the box (page-locked memory): what bounds movi_pml_host from below
import time, torch
dev = torch.device("cuda", 0)
for mb in (9, 37, 150, 300):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory(); h.fill_(65)
    d = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    for name, src, dst in (("H2D", h, d), ("D2H", d, h)):
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        print("%s %4d MiB: %.1f GB/s" % (name, mb, 10 * (mb << 20) / (time.perf_counter() - t0) / 1e9))
