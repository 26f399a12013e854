import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np, torch
from helpers import boundary_tensors, face_settings
from oracle import oracle
from splatter360_amd import synthetic
from test_gpu_parity import run_hip
dev = torch.device("cuda:0")
face = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cloud = synthetic.uniform_cloud(10_000, seed=3, extent=3.0, scale_range=(0.02, 0.3))
S = face_settings(face, 64, 64)
means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
orc = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs)
f = orc.forward()
h = run_hip(S, means, cov6, shs, opac, dev)
d = np.abs(h["image"] - f["image"])
print("max", d.max(), "mean", d.mean())
nc = h["state"]["n_contrib"][0].astype(np.uint32)
bad = np.argwhere(d.max(0) > 1e-5)
print("bad pixels", len(bad), "n_contrib mismatches", (nc != f["n_contrib"]).sum())
for (y, x) in bad[:10]:
    print(y, x, d[:, y, x], "nc hip", nc[y, x], "orc", f["n_contrib"][y, x], "T", h["state"]["final_T"][0][y, x], f["final_T"][y, x])
    # replay the pixel on CPU from the oracle's list to find which entries are near threshold
    t = (y // 16) * 4 + (x // 16)
    s, e = f["ranges"][t]
    T = 1.0
    for k in range(s, e):
        g = f["values"][k]
        dx, dy = f["xy"][g] - np.array([x, y], np.float32)
        A, B, C, o = f["conic_opacity"][g]
        power = np.float32(-0.5) * (A * dx * dx + C * dy * dy) - B * dx * dy
        if power > 0: continue
        al = min(0.99, o * np.exp(power))
        rc = h["state"]["rec_c"][0][g]
        dist = max(abs(dx), abs(dy))
        if al >= 1 / 255 - 1e-6:
            flag = "CULLED?" if dist > rc[3] else ""
            if abs(al - 1 / 255) < 1e-5 or flag:
                print("   entry", k - s, "g", g, "alpha", al, "dist", dist, "rcull", rc[3], "radius", rc[2].view(np.int32) if hasattr(rc[2], 'view') else rc[2], flag)

print("---- replay pixel (5,20)")
y, x = 5, 20
t = (y // 16) * 4 + (x // 16)
s, e = f["ranges"][t]
T = np.float32(1.0)
x0, y0 = (x // 16) * 16, (y // 16) * 16
strip = (y % 16) // 4
for k in range(s, e):
    g = f["values"][k]
    dx, dy = f["xy"][g] - np.array([x, y], np.float32)
    A, B, C, o = f["conic_opacity"][g]
    power = np.float32(-0.5) * (A * dx * dx + C * dy * dy) - B * dx * dy
    rc = h["state"]["rec_c"][0][g]
    r = rc[3]
    gx_, gy_ = f["xy"][g]
    xin = not (gx_ + r < x0 or gx_ - r > x0 + 15)
    ys = y0 + 4 * strip
    yin = not (gy_ + r < ys or gy_ - r > ys + 3)
    if power > 0: continue
    al = np.float32(min(0.99, o * np.exp(power)))
    if al < 1 / 255: continue
    tt = T * (1 - al)
    print(k - s, "g", g, "alpha", al, "T->", tt, "cull-pass", xin and yin, "r", r, "xy", gx_, gy_, "o", o)
    if tt < 1e-4:
        print("DONE at", k - s); break
    T = tt
print("---- entries 610..640 unconditional; list len", e - s)
hl = h["state"]["list"][:f["num_rendered"]]
for k in range(s + 610, min(e, s + 640)):
    g = f["values"][k]
    dx, dy = f["xy"][g] - np.array([x, y], np.float32)
    A, B, C, o = f["conic_opacity"][g]
    power = np.float32(-0.5) * (A * dx * dx + C * dy * dy) - B * dx * dy
    ra = h["state"]["rec_a"][0][g]; rb = h["state"]["rec_b"][0][g]
    print(k - s, "g", g, "hipg", hl[k], "power", power, "alpha", o * np.exp(min(power, 0)), "conic", A, B, C, "hip rec", ra, rb[:2])
