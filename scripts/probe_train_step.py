import sys, time, torch
sys.path.insert(0, '.')
import fastervit_amd
import torch.nn.functional as F
def run(amp, batch=64, steps=5, warmup=2, cl=False):
    torch.manual_seed(0)
    model = fastervit_amd.create_model("faster_vit_0_224", drop_path_rate=0.1).cuda().train()
    if cl: model = model.to(memory_format=torch.channels_last)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05)
    scaler = torch.amp.GradScaler("cuda", enabled=amp)
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(batch, 3, 224, 224, generator=g).cuda()
    if cl: x = x.contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 1000, (batch,), generator=g).cuda()
    losses = []
    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            loss = F.cross_entropy(model(x).float(), y)
        scaler.scale(loss).backward()
        scaler.step(opt); scaler.update()
        losses.append(loss.detach())
    for _ in range(warmup): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"amp={amp} channels_last={cl}: {batch*steps/el:.1f} img/s, {el/steps*1e3:.1f} ms/step, loss {float(losses[0]):.4f} -> {float(losses[-1]):.4f}", flush=True)
for amp, cl in ((False, False), (True, False), (True, True), (False, True)):
    try: run(amp, cl=cl)
    except Exception as e: print("amp", amp, "cl", cl, "FAILED", type(e).__name__, str(e)[:300], flush=True)
# where does the time go (amp off)?
from torch.profiler import profile, ProfilerActivity
torch.manual_seed(0)
model = fastervit_amd.create_model("faster_vit_0_224", drop_path_rate=0.1).cuda().train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
x = torch.randn(64, 3, 224, 224).cuda(); y = torch.randint(0, 1000, (64,)).cuda()
for _ in range(2):
    opt.zero_grad(); F.cross_entropy(model(x), y).backward(); opt.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    opt.zero_grad(); F.cross_entropy(model(x), y).backward(); opt.step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=70))
