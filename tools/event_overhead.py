import torch, time
torch.cuda.init(); x = torch.zeros(1<<20, device="cuda"); torch.cuda.synchronize()
pairs=[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
# empty pairs between real kernels
for a,b in pairs:
    x.add_(1.0)
    a.record(); b.record()
torch.cuda.synchronize()
ts=[a.elapsed_time(b)*1e3 for a,b in pairs]
print("empty pair between kernels: mean %.2f us  median %.2f" % (sum(ts)/len(ts), sorted(ts)[len(ts)//2]))
# pair around a tiny kernel
y = torch.zeros(64, device="cuda")
for a,b in pairs:
    x.add_(1.0)
    a.record(); y.add_(1.0); b.record()
torch.cuda.synchronize()
ts=[a.elapsed_time(b)*1e3 for a,b in pairs]
print("pair around a 64-element add: mean %.2f us  median %.2f" % (sum(ts)/len(ts), sorted(ts)[len(ts)//2]))
for a,b in pairs:
    a.record(); x.add_(1.0); b.record()
torch.cuda.synchronize()
ts=[a.elapsed_time(b)*1e3 for a,b in pairs]
print("pair around a 4 MB add (~3 us kernel): mean %.2f us  median %.2f" % (sum(ts)/len(ts), sorted(ts)[len(ts)//2]))
