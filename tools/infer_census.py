"""Per-shape time census of one UNet forward at the headline shape (Python executor, events around every op call)."""
import collections, os, sys, torch
sys.path.insert(0, os.getcwd())
from ldm_image_generator_amd import ops, synth
from ldm_image_generator_amd.unet import UNet
dev = torch.device("cuda:0")
net = UNet(); net.load_state_dict(synth.fill_state_dict(net.state_dict())); net = net.to(dev).eval()
net.native_forward = False
B = 256
x = torch.randn(B, 8, 32, 32, device=dev); t = torch.full((B,), 500, device=dev)
rec = []
def wrap(name, shape_of):
    fn = getattr(ops, name)
    def inner(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        rec.append((name, shape_of(a, k), e0, e1))
        return r
    setattr(ops, name, inner)
def gshape(a, k):
    return (a[1], a[2], a[3], "g%d" % k.get("groups", 1), "gate" if k.get("weights2") is not None else "", "add" if k.get("addend") is not None else "",
            "conv" if k.get("a_mode", 0) else "", "o%d" % k.get("o_mode", 0), "seg%d" % (len(a[4]) if a[4] is not None else 0))
wrap("gemm", gshape)
for nm in ("channelnorm_film", "window_attention", "avgpool2", "stem_nchw", "head_nchw", "sincos_embed", "film"):
    wrap(nm, lambda a, k: ())
with torch.no_grad():
    net(x=x, time=t, condition=None)
    torch.cuda.synchronize()
    rec.clear()
    net(x=x, time=t, condition=None)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, shp, e0, e1 in rec:
    key = (name,) + tuple(shp)
    n, tt = agg.get(key, (0, 0.0))
    agg[key] = (n + 1, tt + e0.elapsed_time(e1))
tot = sum(v[1] for v in agg.values())
print("total %.2f ms in %d calls" % (tot, len(rec)))
for key, (n, tt) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if key[0] == "gemm":
        m, nn, kk = key[1], key[2], key[3]
        g = int(key[4][1:])
        fl = 2.0 * m * nn * kk * g * (2 if key[5] else 1)
        print("%-60s x%-3d %8.3f ms %8.1f us each %7.1f TF" % (" ".join(str(v) for v in key), n, tt, tt / n * 1e3, fl * n / tt / 1e9))
    else:
        print("%-60s x%-3d %8.3f ms %8.1f us each" % (key[0], n, tt, tt / n * 1e3))
