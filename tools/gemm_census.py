"""Per-shape time census of the bf16 GEMM launches of one training step (events around every call)."""
import collections, os, sys, torch
sys.path.insert(0, os.getcwd())
from ldm_image_generator_amd import dist as ldist, ops, synth, train as ltrain
from ldm_image_generator_amd.ddpm import DDPM
from ldm_image_generator_amd.unet import UNet
dev = torch.device("cuda:0")
net = UNet(); net.load_state_dict(synth.fill_state_dict(net.state_dict())); net = net.to(dev).train()
ltrain.set_precision(net, sys.argv[1] if len(sys.argv) > 1 else "bf16")
ddpm = DDPM(model=net)
opt = torch.optim.AdamW(ddpm.parameters(), lr=1e-4, fused=True)
x = torch.randn(128, 8, 64, 64, generator=torch.Generator().manual_seed(0)).to(dev)
ldist.train_step(ddpm, opt, x, 0, 1)
torch.cuda.synchronize()
rec = []
def wrap(name, shape_of):
    fn = getattr(ops, name)
    def inner(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        rec.append((name, shape_of(a, k), e0, e1))
        return r
    setattr(ops, name, inner)
wrap("gemm_bf16", lambda a, k: (a[1], a[2], a[3], str(a[5].dtype)[6:], "add" if k.get("addend") is not None else ""))
wrap("gemm_bf16_gate_fwd", lambda a, k: (a[1], a[2], a[3], "gatefwd", ""))
wrap("gemm_bf16_gate_bwd", lambda a, k: (a[1], a[2], a[3], "gatebwd", ""))
wrap("gemm_tn_bf16", lambda a, k: (a[3], a[4], a[5], "TN s%d" % a[6], ""))
wrap("gemm", lambda a, k: (a[1], a[2], a[3], "f32", ""))
wrap("gemm_tn", lambda a, k: (a[3], a[4], a[5], "TNf32 s%d" % a[6], ""))
ldist.train_step(ddpm, opt, x, 1, 1)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, shp, e0, e1 in rec:
    key = (name,) + tuple(shp)
    n, t = agg.get(key, (0, 0.0))
    agg[key] = (n + 1, t + e0.elapsed_time(e1))
tot = sum(t for _, t in agg.values())
print("total %.2f ms in %d launches" % (tot, len(rec)))
for key, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    m, nn, kk = key[1], key[2], key[3]
    print("%-22s M=%-7d N=%-5d K=%-5d %-10s %-4s x%-3d %7.2f ms  %6.1f us each  %6.0f TF" % (key[0], m, nn, kk, key[4], key[5], n, t, t / n * 1e3, 2.0 * m * nn * kk * (2 if 'gatefwd' in key[4] else 1) * n / t / 1e9))
