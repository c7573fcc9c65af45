"""Raw host -> device copy rate of the bench host (pinned and pageable, 2.03 GB = the samples of a 64 x 180 s batch):
the floor under the host-buffer entry's time.  For DESIGN.md section 4."""
import json, time
import torch
n = 64 * 180 * 44100
dev = torch.empty(n, dtype=torch.float32, device="cuda")
out = {}
for name, host in (("pinned", torch.empty(n, dtype=torch.float32).pin_memory()), ("pageable", torch.empty(n, dtype=torch.float32))):
    host.fill_(1.0)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); dev.copy_(host, non_blocking=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    out[name] = {"ms": round(t * 1e3, 2), "GB_per_s": round(n * 4 / t / 1e9, 1)}
print(json.dumps(out))
