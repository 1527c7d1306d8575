import torch
x = torch.zeros(1024, device="cuda")
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
for rep in range(2):
    torch.cuda.synchronize()
    for a, b in evs:
        x.add_(1.0)          # a preceding kernel, as in the step
        a.record(); b.record()
    torch.cuda.synchronize()
    ts = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
    print("empty pair us: min %.2f median %.2f max %.2f" % (ts[0], ts[len(ts)//2], ts[-1]))
for rep in range(2):
    torch.cuda.synchronize()
    for a, b in evs:
        x.add_(1.0)
        a.record(); x.add_(1.0); b.record()
    torch.cuda.synchronize()
    ts = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
    print("pair around a 1024-element add us: min %.2f median %.2f max %.2f" % (ts[0], ts[len(ts)//2], ts[-1]))
