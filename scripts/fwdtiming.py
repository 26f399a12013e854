"""Per-wave timing of the forward composite (build with S360_HIPCC_EXTRA=-DS360_DBG_TIMING): which (tile, quadrant) waves of
k_render — and, with S360_FLAG_SPLIT_LISTS, which segment waves of k_render_tail — form its critical path on a given cloud?
usage: fwdtiming.py [encoder_like|surface_like|uniform] [split 0|1]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from splatter360_amd import decoder, rasterizer, synthetic
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "surface_like"
rasterizer.SPLIT_LONG_LISTS = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
cloud = {"encoder_like": lambda: synthetic.encoder_like_cloud(512, 1024), "surface_like": lambda: synthetic.surface_like_cloud(512, 1024),
         "uniform": lambda: synthetic.uniform_cloud(1 << 20, seed=0, extent=5.0)}[name]()
g = [torch.tensor(cloud[k], device=dev).requires_grad_(True) for k in ("means", "covariances", "harmonics", "opacities")]
ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
for _ in range(3):
    faces = decoder.render_views_fused(ext, K, near, far, (256, 256), torch.zeros(3, device=dev), *g, shared_campos=True)
    st = rasterizer.last_state()
torch.cuda.synchronize()
nu = 1536 * 4
t = st.tensors()
ts = t["tile_start"].cpu().numpy().astype(np.int64)
nchunks = int(st._arr(st.layout.chunk_start, 1537, torch.int32)[1536].item())
nseg = 1 if rasterizer.SPLIT_LONG_LISTS else 0
d = st._arr(st.layout.keys_alt, 4 * nu, torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
tl = np.repeat(ts[1:] - ts[:-1], 4)
sl = st._arr(st.layout.strip_last, nu, torch.int32).cpu().numpy()
sc = st._arr(st.layout.surv_count, nu, torch.int32).cpu().numpy()
flag = t["seg_flag"].cpu().numpy()
t0, dur = d[0:4 * nu:4], d[1:4 * nu:4] / 100.0
origin = t0.min()
start = ((t0 - origin) & 0xFFFFFFFF) / 100.0
end = start + dur
print(name, "split", rasterizer.SPLIT_LONG_LISTS, "k_render span us", round(end.max(), 1), " sum of durations / 6144 slots:", round(dur.sum() / 6144, 1), "p99", round(np.percentile(dur, 99), 1),
      "max", dur.max(), " split quadrants", int((flag == 1).sum()))
o = np.argsort(-dur)[:14]
for i in o:
    print("unit", i, "face", i // 1024, "tile", (i // 4) % 256, "q", i % 4, "start", round(start[i], 1), "dur", round(dur[i], 1), "tile_len", tl[i],
          "replay_len", sl[i], "surv_in_front", sc[i], "split", int(flag[i]))
print("units that walk their whole list:", int(((sl >= tl - 64) & (flag != 1)).sum()), "of", nu)
tsx = np.linspace(0, end.max(), 13)
print("running waves at t:", [(round(x), int(((start <= x) & (end > x)).sum())) for x in tsx])
if nseg:
    nwork = int(st.header()[6].item())
    d = st._arr(st.layout.keys_alt, 4 * (nu + nwork), torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    p2, tag = d[4 * nu + 2::4] / 100.0, d[4 * nu + 3::4]
    if nwork:
        comb = (tag >> 31) & 1
        print("k_render_tail (second launch; k_render's span above does not include its phase-1 worker workgroups): segment work items", nwork,
              "| segment composite mean/max us", round(p2[comb == 0].mean(), 1), round(p2[comb == 0].max(), 1),
              "| combining waves", int(comb.sum()), "segment + combine mean/max", round(p2[comb == 1].mean(), 1), round(p2[comb == 1].max(), 1),
              "| longest item / mean item:", round(p2.max() / p2.mean(), 2))
        b0 = d[4 * nu + 0::4]
        st0 = ((b0 - b0.min()) & 0xFFFFFFFF) / 100.0
        rel = ((b0 - origin) & 0xFFFFFFFF) / 100.0      # round 6: phase 2 runs inside k_render's launch — starts on k_render's own clock
        print("  item starts after k_render's first wave: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f | last item end %.1f | heads (split units) end: p50 %.1f p90 %.1f max %.1f"
              % (rel.min(), np.percentile(rel, 10), np.percentile(rel, 50), np.percentile(rel, 90), rel.max(), (rel + p2).max(),
                 np.percentile(end[flag == 1], 50), np.percentile(end[flag == 1], 90), end[flag == 1].max()))
        print("  unsplit units end: p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(end[flag != 1], q) for q in (50, 90, 99, 100)))
        p2own = d[4 * nu + 1::4] / 100.0
        print("  tail span us (first item start -> last item end)", round((st0 + p2).max(), 1), "| item starts: p50", round(np.percentile(st0, 50), 1), "p90",
              round(np.percentile(st0, 90), 1), "max", round(st0.max(), 1), "| own segment (to arrival) mean/p90/max", round(p2own.mean(), 1),
              round(np.percentile(p2own, 90), 1), round(p2own.max(), 1), "| combine part of combiners mean/max",
              round((p2 - p2own)[comb == 1].mean(), 1), round((p2 - p2own)[comb == 1].max(), 1))
        kk = (tag >> 2) & 0x3FF
        for lo, hi in ((2, 4), (4, 8), (8, 16), (16, 64)):
            m = (kk >= lo) & (kk < hi)
            if m.any():
                print("  segments k in [%d,%d): %d items, own-segment mean us %.1f" % (lo, hi, int(m.sum()), p2own[m].mean()))
        for i in np.argsort(-p2)[:6]:
            print("  item tile", int((tag[i] >> 12) & 0x7FFFF), "k", int((tag[i] >> 2) & 0x3FF), "q", int(tag[i] & 3), "us", round(p2[i], 1), "combined", int(comb[i]))

if nseg:
    # round 6: phase-1 units and the phase-2 worker waves' own start stamps (behind the item records)
    ns = int(st.prm.max_segments) if int(st.prm.max_segments) else 8 * (int(st.prm.max_instances) // 2048 + 1)
    base1 = 4 * (nu + 4 * ns)
    d1 = st._arr(st.layout.keys_alt, base1 + 16 * ns, torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    p1 = d1[base1:].reshape(-1, 4)
    ok = p1[:, 2] > 0
    ok &= (((p1[:, 1] - origin) & 0xFFFFFFFF) / 100.0) < 1000.0
    if ok.any():
        wg0 = ((p1[ok, 0] - origin) & 0xFFFFFFFF) / 100.0
        u0 = ((p1[ok, 1] - origin) & 0xFFFFFFFF) / 100.0
        du = p1[ok, 2] / 100.0
        pr = lambda x: " ".join("%.1f" % np.percentile(x, q) for q in (0, 10, 50, 90, 100))
        print("phase-1 (unit, quadrant) records:", int(ok.sum()), "| worker workgroup start (min p10 p50 p90 max):", pr(wg0), "| unit start:", pr(u0),
              "| unit duration:", pr(du), "| last unit end %.1f" % (u0 + du).max())
    base2 = 4 * (nu + 8 * ns)
    d2 = st._arr(st.layout.keys_alt, base2 + 4096, torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    w = ((d2[base2:] - origin) & 0xFFFFFFFF) / 100.0
    w = w[w < 1000.0]
    if w.size:
        print("phase-2 worker waves: %d started, at (min p10 p50 p90 max):" % w.size, " ".join("%.1f" % np.percentile(w, q) for q in (0, 10, 50, 90, 100)))
