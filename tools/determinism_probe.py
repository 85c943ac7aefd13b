"""run-to-run determinism of the eager step on rotating shapes (same process, fresh model each time)"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.argv = [sys.argv[0], "1"]
import runpy
ns = runpy.run_path("/root/repo/tools/eager_shapes_bench.py")
B, pkg, build_model, build_optimizer, DataParallel = ns["B"], ns["pkg"], ns["build_model"], ns["build_optimizer"], ns["DataParallel"]
batches, cfg = ns["batches"], ns["cfg"]
PREFETCH = os.environ.get("PREFETCH", "1") == "1"
PIPE = os.environ.get("PIPE", "1") == "1"
def run(nsteps=int(os.environ.get("NSTEPS", "24")), nshapes=int(os.environ.get("NSHAPES", "5"))):
    torch.manual_seed(1)
    model = build_model(cfg); B.init_weights(model, seed=0); model.train()
    opt = build_optimizer(cfg, model); dp = DataParallel(model)
    if PIPE: opt.enable_pipelined(dp)
    out = []
    for i in range(nsteps):
        losses = model(batches[i % nshapes])
        if PREFETCH: model.prefetch_features(batches[(i + 1) % nshapes])
        if os.environ.get("DIRECT", "0") != "1" or not model.backward_losses(1.0):
            sum(losses.values()).backward()
        dp.finish(); opt.step(dp.grad_scale); opt.zero_grad()
        out.append([float(v.detach()) for v in losses.values()])
    torch.cuda.synchronize()
    w = model.roi_heads.box_head.fc1.weight.detach()
    return out, float(w.double().sum()), float(w.double().abs().sum())
a = run(); b = run(); c = run()
print("prefetch", PREFETCH, "pipelined", PIPE)
for i, (x, y, z) in enumerate(zip(a[0], b[0], c[0])):
    if x != y or x != z:
        print("first differing step", i, x, y, z); break
else:
    print("all", len(a[0]), "steps identical losses")
print("fc1 sums", a[1:], b[1:], c[1:])
