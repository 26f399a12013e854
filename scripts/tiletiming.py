"""Per-wave timing of the forward composite (build with S360_HIPCC_EXTRA=-DS360_DBG_TIMING): how far is the kernel
from a perfectly balanced schedule, and is one sequential chain the bound?"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from splatter360_amd import decoder, rasterizer, synthetic
dev = torch.device("cuda:0")
cloud = synthetic.encoder_like_cloud(512, 1024)
g = [torch.tensor(cloud[k], device=dev) for k in ("means", "covariances", "harmonics", "opacities")]
for _ in range(3):
    faces = decoder.render_cube_faces(torch.eye(4, device=dev), torch.tensor(0.1, device=dev), torch.tensor(10.0, device=dev), 256, torch.zeros(3, device=dev), *g)
torch.cuda.synchronize()
st = rasterizer.last_state()
nt = 1536
d = st._arr(st.layout.keys_alt, 4 * nt * 4, torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
t0, dur, hwid, xcc = d[0::4], d[1::4], d[2::4], d[3::4]
tick = 100.0  # wall_clock64: 100 MHz
start = ((t0 - t0.min()) & 0xFFFFFFFF) / tick
dur = dur / tick
end = start + dur
print("waves", len(dur), "kernel span us", end.max(), "wave dur mean/median/p99/max", dur.mean(), np.median(dur), np.percentile(dur, 99), dur.max())
print("sum of wave durations / (1024 SIMDs) us:", dur.sum() / 1024, " (waves time-share a SIMD: a lower bound of the busy time is below this)")
print("start time pct [50,90,99,max]", np.percentile(start, [50, 90, 99, 100]))
n = np.diff(st.tensors()["tile_start"].cpu().numpy().astype(np.int64))
o = np.argsort(-dur)[:10]
for i in o:
    print("tile", i // 4, "quad", i % 4, "face", i // 4 // 256, "dur", round(dur[i], 1), "start", round(start[i], 1), "tile n", n[i // 4])
h, e = np.histogram(dur, bins=12)
print("hist", h, np.round(e, 1))

simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 15; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7; xc = xcc & 15
key = (((xc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
u, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
print("distinct SIMDs", len(u), "waves per SIMD min/max", cnt.min(), cnt.max(), "xcc values", np.unique(xc), "se", np.unique(se), "sh", np.unique(sh), "cu", np.unique(cu))
fin = np.zeros(len(u)); np.maximum.at(fin, inv, end)
print("per-SIMD finish time us: mean", fin.mean(), "median", np.median(fin), "p10", np.percentile(fin, 10), "max", fin.max())
# mapping of block index -> SIMD bin: how are consecutive blocks placed?
blk = np.arange(len(key)) // 4
order = st._arr(st.layout.tile_order, nt, torch.int32).cpu().numpy()
pos_of_tile = np.empty(nt, np.int64); pos_of_tile[order] = np.arange(nt)
cu_key = key // 4
first = cu_key[0::4]  # CU of each tile (wave 0)
byslot = first[order]  # CU key in dispatch order
print("CU keys of the first 24 dispatched blocks:", byslot[:24])
print("distinct CUs among dispatch slots 0..255:", len(np.unique(byslot[:256])), " 256..511:", len(np.unique(byslot[256:512])))
